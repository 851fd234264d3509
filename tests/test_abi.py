"""CPU suite: libivr_hip.so loads and exports every symbol include/ivr_api.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT
from ivr_amd import _ffi

HEADER = os.path.join(ROOT, "include", "ivr_api.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ivr_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert declared_functions() == sorted(_ffi.EXPORTS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_ffi.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in ivr_api.h but not exported"
    assert _ffi.load().ivr_api_version() == _ffi.API_VERSION


def test_error_slot_without_gpu():
    lib = _ffi.load()
    # NULL arguments are rejected before any HIP call; the message is retrievable per thread
    assert lib.ivr_index_reset(None) == -1
    assert b"NULL" in lib.ivr_last_error(None)
