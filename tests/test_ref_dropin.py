"""The drop-in surface: libdil256_ref.so exports the reference's C++ signatures
(include/dil256_ref.hpp); our mains for the reference's two C++ tests run against it on the GPU."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")

# Itanium-mangled names of the reference's functions (what a TU built against its headers imports)
MANGLED = ["_Z3nttPi", "_Z6invnttPi", "_Z17pointwise_barrettPiPKiS1_", "_Z10ntt2x2_refPi", "_Z13invntt2x2_refPi",
           "_Z13ntt2x2_fwdnttP4BRAMIiE9OPERATION7MAPPING", "_Z13ntt2x2_invnttP4BRAMIiE9OPERATION7MAPPING",
           "_Z10ntt2x2_mulP4BRAMIiEPKS0_7MAPPING", "_Z15resolve_address7MAPPINGj", "_Z7reshapeP4BRAMIiEPKi",
           # the rest of the H6 helper surface (ram_util.h:29-33, util.h:40-50)
           "_Z8read_ramPiPK4BRAMIiEj", "_Z9write_ramP4BRAMIiEjPKi", "_Z19get_twiddle_factorsPiii9OPERATION",
           "_Z13compare_arrayPiS_i", "_Z18compare_bram_arrayP4BRAMIiEPiPKc7MAPPINGi",
           "_Z20print_reshaped_arrayP4BRAMIiEiPKc", "_Z26print_index_reshaped_arrayP4BRAMIiEi"]


def _build():
    import dilithium_amd
    dilithium_amd.load()
    from oracle import oracle as orc
    orc.build()
    subprocess.check_call(["make", "-C", CPP], stdout=subprocess.DEVNULL)


def test_ref_library_exports_reference_symbols(oracle):
    """same mangled names as the compiled reference (cf. oracle/oracle.py Reference.SYMS)"""
    _build()
    from dilithium_amd import _build as b
    lib = C.CDLL(b.REF_LIB)
    for name in MANGLED:
        assert hasattr(lib, name), name
    z = np.ctypeslib.as_array((C.c_int32 * 256).in_dll(lib, "zetas_barrett"))
    assert (z == oracle.zetas()).all()
    # consts_hw.h:7 -- the butterfly unit's twiddle ROM: (z[k], z[2k], z[2k+1]) for k = 1; 4..7; 16..31; 64..127
    hw = np.ctypeslib.as_array((C.c_int32 * (85 * 3)).in_dll(lib, "zetas_barrett_hw")).reshape(85, 3)
    ks = [1] + list(range(4, 8)) + list(range(16, 32)) + list(range(64, 128))
    assert (hw == np.array([[z[k], z[2 * k], z[2 * k + 1]] for k in ks])).all()
    from oracle import oracle as orc
    ref_so = os.path.join(os.path.dirname(orc.__file__), "_ref", "libref.so")
    if os.path.exists(ref_so):     # the compiled reference's own table, when it has been built
        ref_hw = np.ctypeslib.as_array((C.c_int32 * (85 * 3)).in_dll(C.CDLL(ref_so), "zetas_barrett_hw")).reshape(85, 3)
        assert (hw == ref_hw).all()
    ra = lib._Z15resolve_address7MAPPINGj
    ra.restype = C.c_uint
    for m in range(3):
        for a in range(64):
            assert ra(m, a) == oracle.lib.orc_resolve_address(m, a)


def test_header_alone_gives_print_array(tmp_path):
    """util.h:31-40's header template comes with dil256_ref.hpp itself: a TU that includes nothing of the reference compiles,
    and the line it prints is the reference's ("<label> :" + "%3u, " per entry)"""
    src = tmp_path / "pa.cpp"
    src.write_text('#include "dil256_ref.hpp"\nint main() { data_t a[5] = {1, 22, 333, 4444, 8380416}; unsigned char b[2] = {7, 255};\n'
                   '  print_array(a, 5, "r_gold"); print_array(b, 2, "u8"); return 0; }\n')
    exe = tmp_path / "pa"
    subprocess.check_call(["g++", "-O1", "-Wall", "-Wno-format", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode()
    assert out == "r_gold :  1,  22, 333, 4444, 8380416, \nu8 :  7, 255, \n"


def _ref_lib():
    from oracle import oracle as orc
    p = os.path.join(os.path.dirname(orc.__file__), "_ref", "libref.so")
    return C.CDLL(p) if os.path.exists(p) else None


def test_h6_helpers_match_compiled_reference():
    """read_ram / write_ram / get_twiddle_factors / compare_array / compare_bram_array of the drop-in == the
    reference's own (hardware_code/ram_util.cpp:28-94, util.cpp:85-140) on every argument combination"""
    _build()
    ref = _ref_lib()
    if ref is None:
        pytest.skip("compiled reference (oracle/_ref/libref.so) not present")
    from dilithium_amd import _build as b
    ours = C.CDLL(b.REF_LIB)
    rng = np.random.default_rng(5)
    # get_twiddle_factors: every (i, level, mode), incl. MUL_MODE (the reference's default branch)
    for mode in (0, 1, 2):
        for level in (0, 2, 4, 6):
            for i in range(64):
                a, r = (C.c_int32 * 4)(), (C.c_int32 * 4)()
                ours._Z19get_twiddle_factorsPiii9OPERATION(a, i, level, mode)
                ref._Z19get_twiddle_factorsPiii9OPERATION(r, i, level, mode)
                assert list(a) == list(r), (mode, level, i)
    # read_ram / write_ram
    ram = rng.integers(0, 8380417, (64, 4)).astype(np.int32)
    for lib in (ours, ref):
        out = (C.c_int32 * 4)()
        lib._Z8read_ramPiPK4BRAMIiEj(out, ram.ctypes.data_as(C.c_void_p), 37)
        assert list(out) == ram[37].tolist()
        w = ram.copy()
        row = (C.c_int32 * 4)(1, 2, 3, 4)
        lib._Z9write_ramP4BRAMIiEjPKi(w.ctypes.data_as(C.c_void_p), 11, row)
        assert w[11].tolist() == [1, 2, 3, 4] and (np.delete(w, 11, 0) == np.delete(ram, 11, 0)).all()
    # compare_array / compare_bram_array: equal, congruent-but-different representatives, and one corrupted coefficient,
    # under each mapping (stdout of the mismatch report is not compared, only the verdict)
    ra = ours._Z15resolve_address7MAPPINGj
    ra.restype = C.c_uint
    poly = rng.integers(0, 8380417, 256).astype(np.int32)
    for mapping in range(3):
        bram = np.zeros((64, 4), np.int32)
        for r_ in range(64):
            bram[ra(mapping, r_)] = poly[4 * r_:4 * r_ + 4]
        shifted = poly.copy()
        shifted[::3] -= 8380417                      # same residues, negative representatives
        bad = poly.copy()
        bad[129] ^= 1
        for lib in (ours, ref):
            f = lib._Z18compare_bram_arrayP4BRAMIiEPiPKc7MAPPINGi
            p_ = lambda x: x.ctypes.data_as(C.c_void_p)  # noqa: E731
            assert f(p_(bram), p_(poly.copy()), b"t", mapping, 0) == 0
            assert f(p_(bram), p_(shifted.copy()), b"t", mapping, 0) == 0
            assert f(p_(bram), p_(bad.copy()), b"t", mapping, 0) == 1
    for lib in (ours, ref):
        f = lib._Z13compare_arrayPiS_i
        x, y = poly.copy(), poly.copy()
        assert f(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), 256) == 0
        y[255] += 1
        assert f(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), 256) == 1
        assert f(x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), 255) == 0


def test_reference_unchanged_mains_link_against_dropin():
    """ref_test_ntt_ntt2x2.cpp and ntt2x2_test.cpp, UNCHANGED and from /root/reference, link against libdil256_ref.so
    with no reference source on the link line (oracle/Makefile `dropin_mains`).  Build container only."""
    if not os.path.isdir("/root/reference/dilithium-256"):
        pytest.skip("reference tree not present (GPU box)")
    _build()
    from oracle import oracle as orc
    assert orc.build_dropin_mains(force=True)
    out = subprocess.run(["nm", "-u", os.path.join(os.path.dirname(orc.__file__), "_ref", "ntt2x2_test_dropin")],
                         capture_output=True, text=True).stdout
    for sym in ("_Z18compare_bram_arrayP4BRAMIiEPiPKc7MAPPINGi", "_Z13ntt2x2_fwdnttP4BRAMIiE9OPERATION7MAPPING", "_Z3nttPi"):
        assert sym in out, sym        # resolved at run time by the drop-in, not by reference objects


@pytest.mark.gpu
def test_reference_unchanged_main_runs_on_gpu(gpu):
    """the reference's own ref_test_ntt_ntt2x2 main (100 000 + 100 000 iterations, every call a batch of one on the
    GPU) prints OK twice.  The binary is built in the build container (it needs /root/reference) and travels."""
    from oracle import oracle as orc
    exe = os.path.join(os.path.dirname(orc.__file__), "_ref", "ref_test_ntt_ntt2x2_dropin")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_test_ntt_ntt2x2_dropin not built (needs /root/reference)")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("OK") == 2


@pytest.mark.gpu
def test_batched_differential_at_reference_iteration_counts(gpu):
    """10^5 + 10^5 (ref_test_ntt_ntt2x2.cpp:29) and 10^6 x {MUL, NTT, INVNTT, polymul} (ntt2x2_test.cpp:139) through the
    batched host-pointer C-ABI vs the CPU oracle"""
    _build()
    out = subprocess.run([os.path.join(CPP, "test_batched_differential")], capture_output=True, text=True, timeout=1800)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("OK") == 3 and "ERROR" not in out.stdout


# (our re-written analogue of the hardware test main, tests/cpp/test_ntt2x2_hw.cpp, is gone: the reference's UNCHANGED ntt2x2_test.cpp runs against the drop-in
#  at its own 10^6 iterations, tests/test_gpu_mailbox.py::test_reference_unchanged_hw_main_at_its_own_iteration_count)


@pytest.mark.gpu
@pytest.mark.parametrize("level", [2, 3, 5])
def test_cpp_host_program_on_the_cabi(gpu, level):
    """tests/cpp/test_cabi_scheme.cpp: a C++ host (HIP runtime + include/dil256.h only) generating a key, signing and
    verifying ragged messages on a stream, through mu, the multi-GPU host layer and the options"""
    _build()
    out = subprocess.run([os.path.join(CPP, "test_cabi_scheme"), str(level), "300"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr
