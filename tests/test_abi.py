"""CPU-side checks of the C-ABI boundary: the library loads and exports every symbol the
header declares (no compute calls without a GPU)."""
import os

from msml_amd import _lib


def test_header_symbols_exported():
    protos = _lib.parse_header()
    assert "msml_fm_fuse_fwd" in protos and "msml_version" in protos
    lib = _lib.load()                      # raises if any declared symbol is missing
    assert lib.msml_version() == 1
    for name in protos:
        assert hasattr(lib, name), name


def test_header_cites_reference():
    """Every entry point documents the reference call site it replaces."""
    src = open(_lib.HEADER).read()
    assert src.count(".py:") >= 3


def test_status_codes_without_gpu():
    """Argument validation happens before any launch, so it can be exercised on CPU."""
    lib = _lib.load()
    rc = lib.msml_fm_fuse_fwd(None, None, None, 7, 0, 0, 0, None)   # n not a multiple of 8
    assert rc == -1
    assert b"multiple of 8" in lib.msml_last_error()


def test_fast_call_binding_covers_the_header():
    """The generated CPython binding (msml_amd/fastabi_gen.py -> msml_amd/_msml_fastabi.so, built by __graft_entry__.build)
    has a wrapper for every status / count returning prototype of the header, resolves them in the SAME dlopen handle
    ctypes holds, converts None / int / float / objects with data_ptr(), reports a wrong argument count, and hands an
    argument it does not take (a ctypes object) back to the ctypes path of _lib.call."""
    import ctypes
    import __graft_entry__ as ge
    assert ge.build_fastabi()
    fa = _lib._fastabi()
    assert fa, "msml_amd/_msml_fastabi.so did not load"
    protos = _lib.parse_header()
    want = [n for n, (ret, _) in protos.items() if ret in (ctypes.c_int, ctypes.c_long)]
    assert len(want) >= 90 and all(hasattr(fa, n) for n in want)
    lib = _lib.load()
    # same answers as ctypes on pure queries (no GPU): shape queries and validation errors
    args = (128, 128, 256, 28, 28, 28, 28, 3, 3, 1, 1, 1)
    assert fa.msml_conv2d_bnin_acc_applies(*args) == lib.msml_conv2d_bnin_acc_applies(*args)
    assert fa.msml_bn_stats_rows(100000, 64) == lib.msml_bn_stats_rows(100000, 64)
    assert fa.msml_fm_fuse_fwd(None, None, None, 7, 0, 0, 0, None) == -1 and b"multiple of 8" in lib.msml_last_error()

    class T:                                  # what a tensor looks like to the wrapper
        def data_ptr(self):
            return 0
    assert fa.msml_fm_fuse_fwd(T(), T(), T(), 7, 0, 0, 0, 0) == -1
    try:
        fa.msml_bn_stats_rows(1)
        raise AssertionError("argument count not checked")
    except TypeError as e:
        assert "expects 2 arguments" in str(e)
    # _lib.call: an argument the wrapper cannot convert drops that entry to ctypes, which takes it
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert _lib.call("msml_iblock_fwd_tables", ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    assert a.value > 0 and _lib._FAST["msml_iblock_fwd_tables"][4] is None
