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
