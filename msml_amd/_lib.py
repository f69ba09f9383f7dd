"""ctypes binding of libmsml_hip.so, generated from include/msml_hip.h.

The header is the single source of truth: every prototype in it is parsed and bound, so a
symbol declared there but missing from the library fails at load time (loudly), and the
product has no other compute path -- there is no CPU or torch fallback.
"""
import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "msml_hip.h")
LIBPATH = os.environ.get("MSML_LIB", os.path.join(_HERE, "libmsml_hip.so"))   # (override: kernel ablation builds)

F32, BF16 = 0, 1
UNSUPPORTED = -4          # MSML_ERR_UNSUPPORTED
TORCH_DTYPE = {F32: torch.float32, BF16: torch.bfloat16}
DTYPE_OF = {torch.float32: F32, torch.bfloat16: BF16}

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
    "double": ctypes.c_double, "size_t": ctypes.c_size_t,
}
_lib = None
_protos = None


def parse_header(path=HEADER):
    """Return {name: (restype, [(ctype, param_name), ...])} for every prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"\b(int|long|const char\s*\*)\s+(msml_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    params.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, pn = a.rsplit(" ", 1)
                    params.append((_CT[ty.replace("const ", "").strip()], pn))
        rt = {"int": ctypes.c_int, "long": ctypes.c_long}.get(ret, ctypes.c_char_p)
        protos[name] = (rt, params)
    return protos


def load():
    """dlopen the library and bind every header prototype.  Needs no GPU."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIBPATH):
        raise RuntimeError(
            "libmsml_hip.so is missing (%s): run `python __graft_entry__.py` to build it. "
            "msml_amd has no fallback compute path." % LIBPATH)
    lib = ctypes.CDLL(LIBPATH)
    protos = parse_header()
    for name, (ret, params) in protos.items():
        fn = getattr(lib, name, None)
        if fn is None:
            raise RuntimeError("libmsml_hip.so does not export %s declared in msml_hip.h" % name)
        fn.restype = ret
        fn.argtypes = [p[0] for p in params]
    _lib, _protos = lib, protos
    return lib


def exported_symbols():
    load()
    return sorted(_protos)


_DEV = None


def _device():
    global _DEV
    if _DEV is None:
        _DEV = torch.cuda.current_device()      # one process per GPU: fixed for the process
    return _DEV


def raw_stream():
    """hipStream_t of torch's current stream.  torch.cuda.current_stream() costs ~9 us of Python
    per call (device-index resolution); with ~1500 launches per step that was 13 ms of the 29 ms
    host issue time."""
    return torch._C._cuda_getCurrentRawStream(_device())


def current_stream():
    """torch.cuda.current_stream() without the device-index resolution."""
    sid, didx, dtype = torch._C._cuda_getCurrentStream(_device())
    return torch.cuda.Stream(stream_id=sid, device_index=didx, device_type=dtype)


def stream_wait_current(side):
    """`side.wait_stream(<torch's current stream>)` in one library call (msml_stream_wait_stream: no Stream / Event objects
    on the Python side; ~125 forks of the weight-gradient stream per training step)."""
    call("msml_stream_wait_stream", side.cuda_stream, torch._C._cuda_getCurrentRawStream(_DEV if _DEV is not None else _device()))


def _arg(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


_FAST = {}          # name -> [ctypes function, number of parameters, has trailing stream, returns status, C fast-call or None]
_Tensor = torch.Tensor
_FASTABI = None     # the generated CPython binding (msml_amd/fastabi_gen.py), False when it is not there


def _fastabi():
    """msml_amd/_msml_fastabi.so bound to the library handle ctypes holds, or False (MSML_NO_FASTABI=1, not built)."""
    global _FASTABI
    if _FASTABI is None:
        _FASTABI = False
        path = os.path.join(_HERE, "_msml_fastabi.so")
        if not os.environ.get("MSML_NO_FASTABI") and os.path.exists(path):
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location("_msml_fastabi", path)
                mod = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(mod)
                mod.bind(load()._handle)
                _FASTABI = mod
            except Exception as e:                      # stale build (header changed), wrong interpreter: ctypes serves
                import sys
                print("msml_amd: fast-call binding unusable (%r); the ctypes binding serves" % (e,), file=sys.stderr)
    return _FASTABI


def _bind(name):
    lib = load()
    params = _protos[name][1]
    fa = _fastabi()
    ent = [getattr(lib, name), len(params), bool(params) and params[-1][1] == "stream",
           _protos[name][0] is ctypes.c_int, getattr(fa, name, None) if fa else None]
    _FAST[name] = ent
    return ent


def _invoke(name, args):
    """(status or result, returns-a-status) of `name` enqueued on torch's current stream; the trailing `stream` parameter is
    appended automatically."""
    ent = _FAST.get(name)
    if ent is None:
        ent = _bind(name)
    fn, nparams, has_stream, is_status, fast = ent
    if fast is not None:
        # the generated C wrapper converts the arguments itself (None, int, float, anything with data_ptr())
        try:
            if has_stream and len(args) == nparams - 1:
                return fast(*args, torch._C._cuda_getCurrentRawStream(_DEV if _DEV is not None else _device())), is_status
            return fast(*args), is_status
        except (AttributeError, TypeError):
            ent[4] = None            # an argument it does not take (a ctypes array / byref): this entry stays on ctypes
    cargs = [a.data_ptr() if isinstance(a, _Tensor) else a for a in args]
    if has_stream and len(cargs) == nparams - 1:
        cargs.append(torch._C._cuda_getCurrentRawStream(_DEV if _DEV is not None else _device()))
    if len(cargs) != nparams:
        raise TypeError("%s expects %d arguments, got %d" % (name, nparams, len(cargs)))
    return fn(*cargs), is_status


def call(name, *args):
    """Enqueue `name` on torch's current stream; the trailing `stream` parameter is appended
    automatically.  Raises RuntimeError with msml_last_error() on a non-zero status."""
    rc, is_status = _invoke(name, args)
    if rc != 0 and is_status:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, load().msml_last_error().decode()))
    return rc


def try_call(name, *args):
    """Like call(), but returns the status code instead of raising on MSML_ERR_UNSUPPORTED:
    for entry points that cover only part of the shape space and have an unfused alternative."""
    rc, _ = _invoke(name, args)
    if rc not in (0, UNSUPPORTED):
        raise RuntimeError("%s failed (%d): %s" % (name, rc, load().msml_last_error().decode()))
    return rc


def value(name, *args):
    """Call a pure query function (tile sizes, workspace sizes) and return its result."""
    lib = load()
    return getattr(lib, name)(*[_arg(a) for a in args])
