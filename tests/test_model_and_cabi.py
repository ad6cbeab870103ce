"""CPU: (1) the numpy wave model of the kernels' dataflow == oracle, with the twiddle tables
taken from the C++ host library; (2) the C-ABI library loads, exports every symbol that
include/dil256.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle.oracle import Q, splitmix64_polys
from tests.model import wave_model as wm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import dilithium_amd
    return dilithium_amd.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "dil256.h")).read()
    names = set(re.findall(r"\b(dil_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 30
    from dilithium_amd.lib import SIGNATURES
    assert names == set(SIGNATURES), names ^ set(SIGNATURES)
    for n in names:
        assert hasattr(lib, n), f"libdil256.so does not export {n}"


def test_host_tables_match_model(lib):
    f = np.zeros(2048, np.uint32)
    i = np.zeros(2048, np.uint32)
    lib.dil_host_twiddle_tables(f.ctypes.data_as(C.POINTER(C.c_uint32)), i.ctypes.data_as(C.POINTER(C.c_uint32)))
    assert (f.reshape(4, 64, 8)[:, :, :6].transpose(0, 2, 1) == wm.FWD).all()
    assert (i.reshape(4, 64, 8).transpose(0, 2, 1) == wm.INV).all()


def test_host_zetas_match_rom_and_oracle(lib, oracle):
    z = np.zeros(256, np.int32)
    lib.dil_host_zetas(z.ctypes.data_as(C.POINTER(C.c_int32)))
    assert (z == oracle.zetas()).all()
    rom = np.array([int(x, 16) for x in open(os.path.join(ROOT, "tests/golden/zetas_rom.txt")).read().split()])
    assert (np.mod(z.astype(np.int64), Q) == rom).all()


def test_wave_model_matches_oracle(oracle):
    polys = np.concatenate([
        splitmix64_polys(24, seed=3), splitmix64_polys(8, seed=4, lo=-(Q - 1), hi=Q),
        np.array([np.full(256, Q - 1), np.full(256, -(Q - 1)), np.zeros(256), np.arange(256),
                  np.tile([0, Q - 1], 128), np.tile([Q - 1, -(Q - 1)], 128)], dtype=np.int32)])
    good, goodi = oracle.ntt(polys), oracle.invntt(polys)
    for k, a in enumerate(polys):
        assert (wm.ntt_wave(a)[1] == good[k]).all()
        assert (wm.invntt_wave(a)[1] == goodi[k]).all()


def test_no_cpu_fallback_without_gpu(lib):
    """on a box without a GPU every compute entry point must return a HIP error, never a result"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dilithium_amd import api, DilError
    a = np.arange(256, dtype=np.int32)
    before = a.copy()
    with pytest.raises(DilError):
        api.ntt(a)
    assert (a == before).all()
    with pytest.raises(DilError):
        api.init(0)
