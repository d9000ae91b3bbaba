"""CPU (no GPU): liba3r.so loads and exports every symbol include/a3r.h declares; host-only entry points
and argument validation behave; the Python layer fails loudly without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import REPO, rel_err


def _declared_symbols():
    txt = open(os.path.join(REPO, "include", "a3r.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(a3r_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from align3r_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), name
    assert set(_lib.SIGNATURES) == set(declared)       # the ctypes table and the header agree
    assert lib.a3r_version() >= 100


def test_rope_table_host_matches_oracle():
    from align3r_amd import _lib
    from oracle.model_np import rope_tables
    lib = _lib.load()
    cos = np.empty((40, 16), np.float32)
    sin = np.empty((40, 16), np.float32)
    _lib.check(lib.a3r_rope_table_host(cos.ctypes.data_as(C.c_void_p), sin.ctypes.data_as(C.c_void_p), 40, 100.0))
    c, s = rope_tables(40, 100.0, 32)
    assert rel_err(cos, c) < 5e-6 and rel_err(sin, s) < 5e-6    # powf vs np.power: 1 ulp in inv_freq


def test_argument_validation_without_gpu():
    from align3r_amd import _lib
    lib = _lib.load()
    cfg = _lib.ModelConfigC(1000, 24, 16, 768, 12, 12, 4, 16, 100.0, 256, 128, (C.c_int * 4)(96, 192, 384, 768))
    h = C.c_void_p()
    assert lib.a3r_model_create(C.byref(cfg), C.byref(h)) != 0
    assert b"head_dim must be 64" in lib.a3r_last_error()
    with pytest.raises(RuntimeError, match="head_dim"):
        _lib.check(lib.a3r_model_create(C.byref(cfg), C.byref(h)))
    cfg.enc_embed_dim = 1024
    _lib.check(lib.a3r_model_create(C.byref(cfg), C.byref(h)))
    assert lib.a3r_model_workspace_bytes(h, 1, 384, 512) > 2 ** 28     # host-side sizing pass, no device needed
    assert lib.a3r_model_workspace_bytes(h, 1, 380, 512) == 0
    assert lib.a3r_model_packed_bytes(h) > 0
    # forward before finalize is a state error, not a crash
    assert lib.a3r_model_forward(h, *([None] * 4), 1, 64, 64, *([None] * 4), None, 0, None) != 0
    _lib.check(lib.a3r_model_destroy(h))
    assert lib.a3r_align_workspace_bytes(84, 16, 196608) > 0
    assert lib.a3r_linear(None, 32, None, None, 1, 1, 1, 32, None, None) != 0
    assert b"null pointer" in lib.a3r_last_error()


def test_engines_refuse_cpu():
    from align3r_amd.engine import PairEngine
    from align3r_amd.aligner import AlignEngine
    from align3r_amd.weights import TINY
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PairEngine(TINY, {}, device="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        AlignEngine([0], [1], np.zeros((1, 4, 3)), np.zeros((1, 4, 3)), np.zeros((1, 4)), np.zeros((1, 4)), [(2, 2)] * 2, device="cpu")
