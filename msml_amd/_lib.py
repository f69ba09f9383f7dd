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


def _arg(a):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        return a.data_ptr()
    return a


def call(name, *args):
    """Enqueue `name` on torch's current stream; the trailing `stream` parameter is appended
    automatically.  Raises RuntimeError with msml_last_error() on a non-zero status."""
    lib = load()
    fn = getattr(lib, name)
    params = _protos[name][1]
    cargs = [_arg(a) for a in args]
    if params and params[-1][1] == "stream" and len(cargs) == len(params) - 1:
        cargs.append(torch.cuda.current_stream().cuda_stream)
    if len(cargs) != len(params):
        raise TypeError("%s expects %d arguments, got %d" % (name, len(params), len(cargs)))
    rc = fn(*cargs)
    if _protos[name][0] is not ctypes.c_int:
        return rc
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, lib.msml_last_error().decode()))
    return rc


def try_call(name, *args):
    """Like call(), but returns the status code instead of raising on MSML_ERR_UNSUPPORTED:
    for entry points that cover only part of the shape space and have an unfused alternative."""
    lib = load()
    fn = getattr(lib, name)
    params = _protos[name][1]
    cargs = [_arg(a) for a in args]
    if params and params[-1][1] == "stream" and len(cargs) == len(params) - 1:
        cargs.append(torch.cuda.current_stream().cuda_stream)
    if len(cargs) != len(params):
        raise TypeError("%s expects %d arguments, got %d" % (name, len(params), len(cargs)))
    rc = fn(*cargs)
    if rc not in (0, UNSUPPORTED):
        raise RuntimeError("%s failed (%d): %s" % (name, rc, lib.msml_last_error().decode()))
    return rc


def value(name, *args):
    """Call a pure query function (tile sizes, workspace sizes) and return its result."""
    lib = load()
    return getattr(lib, name)(*[_arg(a) for a in args])
