"""The C-ABI library builds for gfx950, loads, and exports every symbol
include/rlhip.h declares (no compute calls: there is no GPU in this tier)."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def libpath():
    from raleigh_amd.build import build_library
    return build_library()


def declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'rlhip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(rlh_[a-z0-9_]+)\s*\(', txt)))


def test_header_and_binding_agree():
    from raleigh_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(libpath):
    dll = ctypes.CDLL(libpath)
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(dll, name), name
    dll.rlh_version.restype = ctypes.c_int
    assert dll.rlh_version() == 100


def test_uninitialised_calls_fail_cleanly(libpath):
    """Before rlh_init every entry point returns an error code and a message."""
    dll = ctypes.CDLL(libpath)
    dll.rlh_last_error.restype = ctypes.c_char_p
    dll.rlh_sync.restype = ctypes.c_int
    assert dll.rlh_sync() != 0
    assert b'rlh_init' in dll.rlh_last_error()
