"""ctypes bindings for the CPU oracle (oracle/dil_oracle.c) and, when it has been
built in the dev container, for the compiled reference (oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
Q = 8380417
N = 256
NATURAL, AFTER_NTT, AFTER_INVNTT = 0, 1, 2

_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


def _p(a, ty=_i32p):
    return a.ctypes.data_as(ty)


def build(force: bool = False) -> None:
    """(Re)build liboracle.so and, if /root/reference is present, _ref/libref.so."""
    so = os.path.join(HERE, "liboracle.so")
    src = os.path.join(HERE, "dil_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref = os.path.join(HERE, "_ref", "libref.so")
    if os.path.isdir("/root/reference/dilithium-256") and (force or not os.path.exists(ref)):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)
    build_dropin_mains(force)


def build_dropin_mains(force: bool = False) -> bool:
    """Link the reference's UNCHANGED test mains (from /root/reference) against the GPU drop-in into _ref/.
    Returns True if both binaries exist afterwards."""
    outs = [os.path.join(HERE, "_ref", n) for n in ("ref_test_ntt_ntt2x2_dropin", "ntt2x2_test_dropin")]
    dropin = os.path.join(os.path.dirname(HERE), "dilithium_amd", "libdil256_ref.so")
    if os.path.isdir("/root/reference/dilithium-256") and os.path.exists(dropin) and \
            (force or not all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(dropin) for o in outs)):
        subprocess.check_call(["make", "-C", HERE, "dropin_mains"], stdout=subprocess.DEVNULL)
    return all(os.path.exists(o) for o in outs)


def _chunks(n, threads):
    """contiguous item ranges for `threads` workers (the C functions are stateless; ctypes drops the GIL)"""
    threads = max(1, min(int(threads), n))
    step = -(-n // threads) if n else 0
    return [(lo, min(lo + step, n)) for lo in range(0, n, step)] if n else []


def _run_chunks(fn, n, threads):
    ch = _chunks(n, threads)
    if len(ch) <= 1:
        for lo, hi in ch:
            fn(lo, hi)
        return
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(len(ch)) as ex:
        list(ex.map(lambda r: fn(*r), ch))


def default_threads():
    try:
        return max(1, min(32, len(os.sched_getaffinity(0))))
    except AttributeError:
        return max(1, min(32, os.cpu_count() or 1))


class Oracle:
    def __init__(self):
        build()
        # DIL_ORACLE_PATH: another build of the same source (the sanitizer build of scripts/san_check.sh)
        self.lib = C.CDLL(os.environ.get("DIL_ORACLE_PATH", os.path.join(HERE, "liboracle.so")))
        L = self.lib
        L.orc_init()
        L.orc_zetas.restype = _i32p
        L.orc_barrett_rtl.restype = C.c_uint32
        L.orc_barrett_rtl.argtypes = [C.c_uint64]
        L.orc_resolve_address.restype = C.c_uint
        L.orc_time_poly_fn.restype = C.c_double
        L.orc_time_poly_fn.argtypes = [C.c_void_p, _i32p, C.c_size_t, C.c_int]
        L.orc_time_verify_core.restype = C.c_double
        for name in ("orc_ntt_batch", "orc_invntt_batch", "orc_canon_batch", "orc_ntt2x2_batch",
                     "orc_invntt2x2_batch"):
            getattr(L, name).argtypes = [_i32p, C.c_size_t]
        L.orc_pointwise_batch.argtypes = [_i32p, _i32p, _i32p, C.c_size_t]
        L.orc_decompose.argtypes = [C.c_int, C.c_int32, _i32p]
        L.orc_decompose_rtl.argtypes = [C.c_int, C.c_int32, _i32p]

    # -- tables ---------------------------------------------------------
    def zetas(self) -> np.ndarray:
        return np.ctypeslib.as_array(self.lib.orc_zetas(), shape=(N,)).copy()

    # -- polynomial batches: arrays [..., 256] int32, returns new arrays ---
    def _batch(self, fn, a, canon=True):
        a = np.ascontiguousarray(a, dtype=np.int32).copy()
        n = a.size // N
        getattr(self.lib, fn)(_p(a), n)
        if canon:
            self.lib.orc_canon_batch(_p(a), n)
        return a

    def ntt(self, a, canon=True):
        return self._batch("orc_ntt_batch", a, canon)

    def invntt(self, a, canon=True):
        return self._batch("orc_invntt_batch", a, canon)

    def ntt2x2(self, a, canon=True):
        return self._batch("orc_ntt2x2_batch", a, canon)

    def invntt2x2(self, a, canon=True):
        return self._batch("orc_invntt2x2_batch", a, canon)

    def pointwise(self, a, b, canon=True):
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        c = np.empty_like(a)
        self.lib.orc_pointwise_batch(_p(c), _p(a), _p(b), a.size // N)
        if canon:
            self.lib.orc_canon_batch(_p(c), a.size // N)
        return c

    # -- bram (hardware-model API) ---------------------------------------
    def bram_fwdntt(self, ram, mapping):
        ram = np.ascontiguousarray(ram, dtype=np.int32).copy()
        for r in ram.reshape(-1, N):
            self.lib.orc_bram_fwdntt(_p(r), int(mapping))
        return ram

    def bram_invntt(self, ram, mapping):
        ram = np.ascontiguousarray(ram, dtype=np.int32).copy()
        for r in ram.reshape(-1, N):
            self.lib.orc_bram_invntt(_p(r), int(mapping))
        return ram

    def bram_mul(self, ram, mul_ram, mapping):
        ram = np.ascontiguousarray(ram, dtype=np.int32).copy()
        mul_ram = np.ascontiguousarray(mul_ram, dtype=np.int32)
        for r, m in zip(ram.reshape(-1, N), mul_ram.reshape(-1, N)):
            self.lib.orc_bram_mul(_p(r), _p(m), int(mapping))
        return ram

    # -- Dilithium pipelines ------------------------------------------------
    def params(self, level):
        class P(C.Structure):
            _fields_ = [(k, C.c_int) for k in ("K", "L", "eta", "tau", "omega", "beta")] + \
                       [("gamma1", C.c_int32), ("gamma2", C.c_int32)]
        p = P()
        if self.lib.orc_get_params(int(level), C.byref(p)):
            raise ValueError(f"bad level {level}")
        return p

    def butterfly_circuit(self, data_in, w, mode):
        """orc_butterfly_circuit on rows of 4 lanes / 4 twiddles: canonical outputs"""
        data_in = np.ascontiguousarray(data_in, dtype=np.int32).reshape(-1, 4)
        w = np.ascontiguousarray(w, dtype=np.int32).reshape(-1, 4)
        out = np.empty_like(data_in)
        for o_, i_, w_ in zip(out, data_in, w):
            self.lib.orc_butterfly_circuit(int(mode), _p(i_), _p(w_), _p(o_))
        return out

    def matvec(self, K, L, A, y, shared_A=False):
        y = np.ascontiguousarray(y, dtype=np.int32)
        A = np.ascontiguousarray(A, dtype=np.int32)
        n = y.size // (L * N)
        w = np.empty((n, K, N), dtype=np.int32)
        y = y.reshape(n, L * N)
        A2 = A.reshape(-1, K * L * N)

        def run(lo, hi):
            self.lib.orc_matvec_batch(K, L, _p(A2[0 if shared_A else lo:]), _p(y[lo:]), _p(w[lo:]), C.c_size_t(hi - lo), int(shared_A))
        _run_chunks(run, n, default_threads() if n >= 512 else 1)
        return w

    def verify_core(self, level, A, z, c, t1, h, shared_pk=False):
        p = self.params(level)
        z = np.ascontiguousarray(z, dtype=np.int32)
        n = z.size // (p.L * N)
        A = np.ascontiguousarray(A, dtype=np.int32)
        c = np.ascontiguousarray(c, dtype=np.int32)
        t1 = np.ascontiguousarray(t1, dtype=np.int32)
        h = np.ascontiguousarray(h, dtype=np.uint8)
        w1 = np.empty((n, p.K, N), dtype=np.uint8)
        z, c, h = z.reshape(n, -1), c.reshape(n, -1), h.reshape(n, -1)
        A2, t2 = A.reshape(-1, p.K * p.L * N), t1.reshape(-1, p.K * N)

        def run(lo, hi):
            k = 0 if shared_pk else lo
            self.lib.orc_verify_core_batch(int(level), _p(A2[k:]), _p(z[lo:]), _p(c[lo:]), _p(t2[k:]), _p(h[lo:], _u8p),
                                           _p(w1[lo:], _u8p), C.c_size_t(hi - lo), int(shared_pk))
        _run_chunks(run, n, default_threads() if n >= 512 else 1)
        return w1

    def time_verify_core(self, level, A, z, c, t1, h, shared_pk=False):
        p = self.params(level)
        n = z.size // (p.L * N)
        w1 = np.empty((n, p.K, N), dtype=np.uint8)
        return self.lib.orc_time_verify_core(int(level), _p(A), _p(z), _p(c), _p(t1), _p(h, _u8p),
                                             _p(w1, _u8p), C.c_size_t(n), int(shared_pk))

    def sign_phase1(self, level, A, y):
        p = self.params(level)
        y = np.ascontiguousarray(y, dtype=np.int32).reshape(-1, p.L, N)
        A = np.ascontiguousarray(A, dtype=np.int32).reshape(-1, p.K, p.L, N)
        n = y.shape[0]
        w1 = np.empty((n, p.K, N), dtype=np.uint8)
        w0 = np.empty((n, p.K, N), dtype=np.int32)
        shared = A.shape[0] == 1

        def run(lo, hi):
            self.lib.orc_sign_phase1_batch(int(level), _p(A[0 if shared else lo:]), _p(y[lo:]), _p(w1[lo:], _u8p), _p(w0[lo:]),
                                           C.c_size_t(hi - lo), int(shared))
        _run_chunks(run, n, default_threads() if n >= 512 else 1)
        return w1, w0

    def sign_phase2(self, level, c, y, w0, w1, s1hat, s2hat, t0hat):
        p = self.params(level)
        y = np.ascontiguousarray(y, dtype=np.int32).reshape(-1, p.L, N)
        n = y.shape[0]
        c = np.ascontiguousarray(c, dtype=np.int32).reshape(n, N)
        w0 = np.ascontiguousarray(w0, dtype=np.int32).reshape(n, p.K, N)
        w1 = np.ascontiguousarray(w1, dtype=np.uint8).reshape(n, p.K, N)
        s1hat = np.ascontiguousarray(s1hat, dtype=np.int32).reshape(-1, p.L, N)
        s2hat = np.ascontiguousarray(s2hat, dtype=np.int32).reshape(-1, p.K, N)
        t0hat = np.ascontiguousarray(t0hat, dtype=np.int32).reshape(-1, p.K, N)
        z = np.empty((n, p.L, N), dtype=np.int32)
        h = np.empty((n, p.K, N), dtype=np.uint8)
        flags = np.empty(n, dtype=np.int32)
        shared = s1hat.shape[0] == 1

        def run(lo, hi):
            k = 0 if shared else lo
            self.lib.orc_sign_phase2_batch(int(level), _p(c[lo:]), _p(y[lo:]), _p(w0[lo:]), _p(w1[lo:], _u8p), _p(s1hat[k:]),
                                           _p(s2hat[k:]), _p(t0hat[k:]), _p(z[lo:]), _p(h[lo:], _u8p), _p(flags[lo:]),
                                           C.c_size_t(hi - lo), int(shared))
        _run_chunks(run, n, default_threads() if n >= 512 else 1)
        return z, h, flags

    # -- timing helpers (cpu_baseline) --------------------------------------
    def time_poly_fn(self, fn_addr, a, reps):
        a = np.ascontiguousarray(a, dtype=np.int32)
        return self.lib.orc_time_poly_fn(C.c_void_p(fn_addr), _p(a), a.size // N, int(reps))

    def fn_addr(self, name):
        return C.cast(getattr(self.lib, name), C.c_void_p).value


class Reference:
    """The reference's own compiled C++ (oracle/_ref/libref.so, built by oracle/Makefile from
    /root/reference/dilithium-256).  Mangled C++ names are bound directly; no shim source."""

    SYMS = {
        "ntt": "_Z3nttPi",
        "invntt": "_Z6invnttPi",
        "pointwise_barrett": "_Z17pointwise_barrettPiPKiS1_",
        "ntt2x2_ref": "_Z10ntt2x2_refPi",
        "invntt2x2_ref": "_Z13invntt2x2_refPi",
        "ntt2x2_fwdntt": "_Z13ntt2x2_fwdnttP4BRAMIiE9OPERATION7MAPPING",
        "ntt2x2_invntt": "_Z13ntt2x2_invnttP4BRAMIiE9OPERATION7MAPPING",
        "ntt2x2_mul": "_Z10ntt2x2_mulP4BRAMIiEPKS0_7MAPPING",
        "resolve_address": "_Z15resolve_address7MAPPINGj",
    }
    FORWARD_NTT_MODE, INVERSE_NTT_MODE, MUL_MODE = 0, 1, 2

    @staticmethod
    def path():
        return os.path.join(HERE, "_ref", "libref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.path())

    def __init__(self):
        if not self.available():
            raise FileNotFoundError("oracle/_ref/libref.so not built (needs /root/reference; run make -C oracle ref)")
        self.lib = C.CDLL(self.path())
        self.f = {k: getattr(self.lib, v) for k, v in self.SYMS.items()}
        self.zetas_barrett = np.ctypeslib.as_array((C.c_int32 * N).in_dll(self.lib, "zetas_barrett")).copy()

    def addr(self, name):
        return C.cast(self.f[name], C.c_void_p).value

    def _each(self, name, a):
        a = np.ascontiguousarray(a, dtype=np.int32).copy()
        fn = self.f[name]
        for r in a.reshape(-1, N):
            fn(_p(r))
        return a

    def ntt(self, a): return self._each("ntt", a)
    def invntt(self, a): return self._each("invntt", a)
    def ntt2x2_ref(self, a): return self._each("ntt2x2_ref", a)
    def invntt2x2_ref(self, a): return self._each("invntt2x2_ref", a)

    def pointwise_barrett(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.int32)
        b = np.ascontiguousarray(b, dtype=np.int32)
        c = np.empty_like(a)
        for cr, ar, br in zip(c.reshape(-1, N), a.reshape(-1, N), b.reshape(-1, N)):
            self.f["pointwise_barrett"](_p(cr), _p(ar), _p(br))
        return c

    def buttefly_circuit(self, data_in, w, mode):
        """the reference's header template buttefly_circuit<data2_t, data_t> (butterfly_unit.h:112-196) through oracle/ref_shim.cpp: rows of
        4 lanes and 4 twiddles -> rows of 4 raw outputs"""
        data_in = np.ascontiguousarray(data_in, dtype=np.int32).reshape(-1, 4)
        w = np.ascontiguousarray(w, dtype=np.int32).reshape(-1, 4)
        out = np.empty_like(data_in)
        for o_, i_, w_ in zip(out, data_in, w):
            self.lib.ref_buttefly_circuit(_p(o_), _p(i_), _p(w_), int(mode))
        return out

    def butterfly(self, mode, zeta, aj, ajlen):
        """butterfly<data2_t, data_t> (butterfly_unit.h:29-110), one call: (bj, bjlen) raw"""
        bj, bl = C.c_int32(), C.c_int32()
        self.lib.ref_butterfly(int(mode), C.byref(bj), C.byref(bl), int(zeta), int(aj), int(ajlen))
        return bj.value, bl.value

    def bram_fwdntt(self, ram, mapping):
        ram = np.ascontiguousarray(ram, dtype=np.int32).copy()
        for r in ram.reshape(-1, N):
            self.f["ntt2x2_fwdntt"](_p(r), self.FORWARD_NTT_MODE, int(mapping))
        return ram

    def bram_invntt(self, ram, mapping):
        ram = np.ascontiguousarray(ram, dtype=np.int32).copy()
        for r in ram.reshape(-1, N):
            self.f["ntt2x2_invntt"](_p(r), self.INVERSE_NTT_MODE, int(mapping))
        return ram

    def bram_mul(self, ram, mul_ram, mapping):
        ram = np.ascontiguousarray(ram, dtype=np.int32).copy()
        mul_ram = np.ascontiguousarray(mul_ram, dtype=np.int32)
        for r, m in zip(ram.reshape(-1, N), mul_ram.reshape(-1, N)):
            self.f["ntt2x2_mul"](_p(r), _p(m), int(mapping))
        return ram


def canon(a):
    """canonical residues in [0, q) of an int array"""
    return np.mod(np.asarray(a, dtype=np.int64), Q).astype(np.int32)


def splitmix64_polys(n, seed=0, lo=0, hi=Q):
    """Portable seeded inputs (the reference uses libc rand()): uniform in [lo, hi)."""
    off = np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(n * N, dtype=np.uint64) + off
    with np.errstate(over="ignore"):
        z = idx * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (lo + (z % np.uint64(hi - lo)).astype(np.int64)).astype(np.int32).reshape(n, N)
