"""Host-side mirror of the reference's dilithium-256/ interface on batches.

Same names, argument meaning and in-place behaviour as the reference functions
(ref_ntt.h:30-36, ref_ntt2x2.h:31-33, ntt2x2.h:30-34), lifted from one `data_t[256]` to a
dense batch `[..., 256]`:

  * torch CUDA int32 tensors  -> device entry points (asynchronous, on torch's current stream)
  * numpy int32 arrays        -> host entry points (synchronous H2D / kernel / D2H)

All results are canonical residues in [0, q).  There is no CPU implementation here: without
the HIP library and a GPU every call raises DilError.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib
from . import LEVELS, N

try:  # torch is plumbing (device memory, streams); numpy-only use does not need it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_tensor(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype=None, name="tensor"):
    if not (t.is_cuda and t.is_contiguous()):
        raise ValueError(f"{name} must be a contiguous CUDA tensor")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} must have dtype {dtype}")
    return C.c_void_p(t.data_ptr())


def _batch(t):
    n = t.numel() if _is_tensor(t) else t.size
    if n % N:
        raise ValueError("size must be a multiple of 256")
    return n // N


def _np(a):
    if not (isinstance(a, np.ndarray) and a.dtype == np.int32 and a.flags.c_contiguous):
        raise ValueError("expected a C-contiguous numpy int32 array")
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def init(device: int = -1) -> None:
    _lib.check(_lib.load().dil_init(device), "dil_init")


def set_option(name: str, value: int) -> None:
    """process-wide option of the library (include/dil256.h): "fused_mode", "fuse_wire", "zeroize", ..."""
    _lib.check(_lib.load().dil_set_option(name.encode(), int(value)), f"dil_set_option({name})")


def get_option(name: str) -> int:
    v = C.c_int()
    _lib.check(_lib.load().dil_get_option(name.encode(), C.byref(v)), f"dil_get_option({name})")
    return int(v.value)


# ---- H2/H3/H5 -----------------------------------------------------------------------------------
def ntt(a):
    """in-place forward NTT of every polynomial of `a` (ref_ntt.cpp:28-47)"""
    L = _lib.load()
    if _is_tensor(a):
        _lib.check(L.dil_ntt_dev(_dev(a, torch.int32, "a"), _batch(a), _stream()), "dil_ntt_dev")
    else:
        _lib.check(L.dil_ntt_host(_np(a), _batch(a)), "dil_ntt_host")
    return a


def invntt(a):
    """in-place inverse NTT incl. the 256^-1 scaling (ref_ntt.cpp:59-87)"""
    L = _lib.load()
    if _is_tensor(a):
        _lib.check(L.dil_invntt_dev(_dev(a, torch.int32, "a"), _batch(a), _stream()), "dil_invntt_dev")
    else:
        _lib.check(L.dil_invntt_host(_np(a), _batch(a)), "dil_invntt_host")
    return a


# the radix-2x2 reference computes the same maps (ref_ntt2x2.cpp:37-82,100-145)
ntt2x2_ref = ntt
invntt2x2_ref = invntt


def pointwise_barrett(c, a, b):
    """c = a o b (ref_ntt.cpp:49-57); c may alias a"""
    L = _lib.load()
    if _is_tensor(a):
        _lib.check(L.dil_pointwise_dev(_dev(c, torch.int32), _dev(a, torch.int32), _dev(b, torch.int32),
                                       _batch(a), _stream()), "dil_pointwise_dev")
    else:
        _lib.check(L.dil_pointwise_host(_np(c), _np(a), _np(b), _batch(a)), "dil_pointwise_host")
    return c


def polymul(c, a, b):
    """c = a * b in Z_q[x] / (x^256 + 1): the reference's polymul chain (ntt, ntt, pointwise_barrett, invntt; ntt2x2_test.cpp:109-137) as ONE
    fused kernel; host arrays: one upload of a and b, one download of c.  c may alias a or b.  |a|, |b| < 2^26."""
    L = _lib.load()
    if _is_tensor(a):
        _lib.check(L.dil_polymul_dev(_dev(c, torch.int32), _dev(a, torch.int32), _dev(b, torch.int32), _batch(a), _stream()), "dil_polymul_dev")
    else:
        _lib.check(L.dil_polymul_host(_np(c), _np(a), _np(b), _batch(a)), "dil_polymul_host")
    return c


def pointwise_acc(c, acc, a, b):
    """c = acc + a o b : the RTL's MULT mode is a multiply-accumulate (butterfly.v:144-150)"""
    _lib.check(_lib.load().dil_pointwise_acc_dev(_dev(c, torch.int32), _dev(acc, torch.int32), _dev(a, torch.int32),
                                                 _dev(b, torch.int32), _batch(a), _stream()), "dil_pointwise_acc_dev")
    return c


def poly_add(c, a, b):
    _lib.check(_lib.load().dil_poly_add_dev(_dev(c, torch.int32), _dev(a, torch.int32), _dev(b, torch.int32),
                                            _batch(a), _stream()), "dil_poly_add_dev")
    return c


def poly_sub(c, a, b):
    _lib.check(_lib.load().dil_poly_sub_dev(_dev(c, torch.int32), _dev(a, torch.int32), _dev(b, torch.int32),
                                            _batch(a), _stream()), "dil_poly_sub_dev")
    return c


# ---- H6: hardware-model API on bram ([..., 64, 4]) ---------------------------------------------
def ntt2x2_fwdntt(ram, mapping):
    L = _lib.load()
    if _is_tensor(ram):
        _lib.check(L.dil_bram_fwdntt_dev(_dev(ram, torch.int32), _batch(ram), int(mapping), _stream()), "dil_bram_fwdntt_dev")
    else:
        _lib.check(L.dil_bram_fwdntt_host(_np(ram), _batch(ram), int(mapping)), "dil_bram_fwdntt_host")
    return ram


def ntt2x2_invntt(ram, mapping):
    L = _lib.load()
    if _is_tensor(ram):
        _lib.check(L.dil_bram_invntt_dev(_dev(ram, torch.int32), _batch(ram), int(mapping), _stream()), "dil_bram_invntt_dev")
    else:
        _lib.check(L.dil_bram_invntt_host(_np(ram), _batch(ram), int(mapping)), "dil_bram_invntt_host")
    return ram


def ntt2x2_mul(ram, mul_ram, mapping):
    L = _lib.load()
    if _is_tensor(ram):
        _lib.check(L.dil_bram_mul_dev(_dev(ram, torch.int32), _dev(mul_ram, torch.int32), _batch(ram), int(mapping),
                                      _stream()), "dil_bram_mul_dev")
    else:
        _lib.check(L.dil_bram_mul_host(_np(ram), _np(mul_ram), _batch(ram), int(mapping)), "dil_bram_mul_host")
    return ram


# ---- H8-H10: fused Dilithium pipelines (device tensors only) -------------------------------------
def _kl(level):
    if level not in LEVELS:
        raise ValueError("level must be 2, 3 or 5")
    return LEVELS[level]


def matvec(A, y, level, shared_A=False, out=None):
    """w = INTT(A o NTT(y)); A [B|1,K,L,256], y [B,L,256] -> w [B,K,256]"""
    K, Lv = _kl(level)
    B = y.numel() // (Lv * N)
    w = out if out is not None else torch.empty((B, K, N), dtype=torch.int32, device=y.device)
    _lib.check(_lib.load().dil_matvec_dev(_dev(w, torch.int32), _dev(A, torch.int32), _dev(y, torch.int32), level, B,
                                          int(shared_A), _stream()), "dil_matvec_dev")
    return w


def verify_core(A, z, c, t1, h, level, shared_pk=False, out=None):
    """w1 = UseHint(h, INTT(A o NTT(z) - NTT(c) o NTT(t1 2^13))) -> uint8 [B,K,256]; numpy operands take the host-pointer entry point"""
    K, Lv = _kl(level)
    if not _is_tensor(z):
        B = z.size // (Lv * N)
        w1 = out if out is not None else np.empty((B, K * N), dtype=np.uint8)
        _lib.check(_lib.load().dil_verify_core_host(_np8(w1), _np(A), _np(z), _np(c), _np(t1), _np8(h.reshape(B, K * N)), level, B, int(shared_pk)),
                   "dil_verify_core_host")
        return w1.reshape(B, K, N)
    B = z.numel() // (Lv * N)
    w1 = out if out is not None else torch.empty((B, K, N), dtype=torch.uint8, device=z.device)
    _lib.check(_lib.load().dil_verify_core_dev(_dev(w1, torch.uint8), _dev(A, torch.int32), _dev(z, torch.int32),
                                               _dev(c, torch.int32), _dev(t1, torch.int32), _dev(h, torch.uint8),
                                               level, B, int(shared_pk), _stream()), "dil_verify_core_dev")
    return w1


def sign_phase1(A, y, level, shared_key=False):
    K, Lv = _kl(level)
    B = y.numel() // (Lv * N)
    w1 = torch.empty((B, K, N), dtype=torch.uint8, device=y.device)
    w0 = torch.empty((B, K, N), dtype=torch.int32, device=y.device)
    _lib.check(_lib.load().dil_sign_phase1_dev(_dev(w1, torch.uint8), _dev(w0, torch.int32), _dev(A, torch.int32),
                                               _dev(y, torch.int32), level, B, int(shared_key), _stream()),
               "dil_sign_phase1_dev")
    return w1, w0


def sign_phase2(c, y, w0, w1, s1hat, s2hat, t0hat, level, shared_key=False, small_key=False):
    """small_key: the caller vouches for a key decoded from secret-key bytes and a challenge from SampleInBall
    (dil_sign_phase2_skey_dev: the small-product kernels); default: any residues (dil_sign_phase2_dev)"""
    K, Lv = _kl(level)
    B = y.numel() // (Lv * N)
    z = torch.empty((B, Lv, N), dtype=torch.int32, device=y.device)
    h = torch.empty((B, K, N), dtype=torch.uint8, device=y.device)
    flags = torch.empty((B,), dtype=torch.int32, device=y.device)
    args = (_dev(z, torch.int32), _dev(h, torch.uint8), _dev(flags, torch.int32), _dev(c, torch.int32), _dev(y, torch.int32),
            _dev(w0, torch.int32), _dev(w1, torch.uint8), _dev(s1hat, torch.int32), _dev(s2hat, torch.int32), _dev(t0hat, torch.int32),
            level, B, int(shared_key))
    if small_key:
        _lib.check(_lib.load().dil_sign_phase2_skey_dev(*args, 0, _stream()), "dil_sign_phase2_skey_dev")
    else:
        _lib.check(_lib.load().dil_sign_phase2_dev(*args, _stream()), "dil_sign_phase2_dev")
    return z, h, flags


def sign_phase2_early(c, y, w0, w1, s1hat, s2hat, t0hat, level, shared_key=False, small_key=False):
    """phase 2 as the signing loop runs it: stops at an attempt's first failed check (r0 rows -> 2, z rows -> 1, c t0
    rows -> 4); z, h complete only where flags == 0.  w0 is IN/OUT (holds r0 of the evaluated rows afterwards).
    small_key: as the loop itself calls it (dil_sign_phase2_skey_dev with early_exit = 1)"""
    K, Lv = _kl(level)
    B = y.numel() // (Lv * N)
    z = torch.zeros((B, Lv, N), dtype=torch.int32, device=y.device)
    h = torch.zeros((B, K, N), dtype=torch.uint8, device=y.device)
    flags = torch.empty((B,), dtype=torch.int32, device=y.device)
    args = (_dev(z, torch.int32), _dev(h, torch.uint8), _dev(flags, torch.int32), _dev(c, torch.int32), _dev(y, torch.int32),
            _dev(w0, torch.int32), _dev(w1, torch.uint8), _dev(s1hat, torch.int32), _dev(s2hat, torch.int32), _dev(t0hat, torch.int32),
            level, B, int(shared_key))
    if small_key:
        _lib.check(_lib.load().dil_sign_phase2_skey_dev(*args, 1, _stream()), "dil_sign_phase2_skey_dev")
    else:
        _lib.check(_lib.load().dil_sign_phase2_early_dev(*args, _stream()), "dil_sign_phase2_early_dev")
    return z, h, flags


def launch_info(family: str) -> dict:
    """most recent launch of a persistent kernel family: grid, items per workgroup and step, batch, launches so far"""
    g, ipb, items, n = C.c_int(), C.c_int(), C.c_size_t(), C.c_size_t()
    _lib.check(_lib.load().dil_launch_info(family.encode(), C.byref(g), C.byref(ipb), C.byref(items), C.byref(n)), "dil_launch_info")
    return {"grid": g.value, "items_per_block": ipb.value, "items": items.value, "launches": n.value,
            "steps": -(-items.value // max(1, g.value * ipb.value))}


# ---- row N1: SHAKE-bound samplers on the device (uint8 / int32 CUDA tensors) -----------------------
def shake256(data, out_bytes):
    """out[i] = SHAKE256(data[i]); data uint8 [B, n] with n % 8 == 0, out_bytes % 8 == 0"""
    B, n = data.shape
    out = torch.empty((B, out_bytes), dtype=torch.uint8, device=data.device)
    _lib.check(_lib.load().dil_shake256_dev(_dev(out, torch.uint8), out_bytes, _dev(data, torch.uint8), n, B, _stream()),
               "dil_shake256_dev")
    return out


def expand_a(rho, level):
    """A [B,K,L,256] from rho uint8 [B,32]"""
    K, Lv = _kl(level)
    B = rho.shape[0]
    A = torch.empty((B, K, Lv, N), dtype=torch.int32, device=rho.device)
    _lib.check(_lib.load().dil_expand_a_dev(_dev(A, torch.int32), _dev(rho, torch.uint8), level, B, _stream()), "dil_expand_a_dev")
    return A


def expand_mask(rhoprime, kappa, level):
    """y [B,L,256] canonical from rho' uint8 [B,64] and int32 nonce base kappa [B]"""
    K, Lv = _kl(level)
    B = rhoprime.shape[0]
    y = torch.empty((B, Lv, N), dtype=torch.int32, device=rhoprime.device)
    _lib.check(_lib.load().dil_expand_mask_dev(_dev(y, torch.int32), _dev(rhoprime, torch.uint8), _dev(kappa, torch.int32),
                                               level, B, _stream()), "dil_expand_mask_dev")
    return y


def sample_in_ball(ctilde, level):
    """c [B,256] (+-1 as 1 / q-1) from c~ uint8 [B,32]"""
    B = ctilde.shape[0]
    c = torch.empty((B, N), dtype=torch.int32, device=ctilde.device)
    _lib.check(_lib.load().dil_sample_in_ball_dev(_dev(c, torch.int32), _dev(ctilde, torch.uint8), level, B, _stream()),
               "dil_sample_in_ball_dev")
    return c


def challenge(mu, w1_packed, level):
    """(c~ uint8 [B,32], c int32 [B,256]) from mu uint8 [B,64] and the packed w1 (pack_w1) -- one launch (gen_c.v is one module)"""
    K, _ = _kl(level)
    B = mu.shape[0]
    assert mu.shape == (B, 64) and w1_packed.shape == (B, K * (192 if level == 2 else 128))
    ct = torch.empty((B, 32), dtype=torch.uint8, device=mu.device)
    c = torch.empty((B, N), dtype=torch.int32, device=mu.device)
    _lib.check(_lib.load().dil_challenge_dev(_dev(ct, torch.uint8), _dev(c, torch.int32), _dev(mu, torch.uint8), _dev(w1_packed, torch.uint8),
                                             level, B, _stream()), "dil_challenge_dev")
    return ct, c


def pack_w1(w1, level):
    """[B,K,256] uint8 -> [B, K*128] (levels 3/5) or [B, K*192] (level 2) uint8"""
    K, _ = _kl(level)
    B = w1.shape[0]
    out = torch.empty((B, K * (192 if level == 2 else 128)), dtype=torch.uint8, device=w1.device)
    _lib.check(_lib.load().dil_pack_w1_dev(_dev(out, torch.uint8), _dev(w1, torch.uint8), level, B, _stream()), "dil_pack_w1_dev")
    return out


# ---- row N3 (first step): whole sequences as one call ----------------------------------------------
def verify(A, ctilde, z, t1, h, mu, level, shared_pk=False):
    """verdict int32 [B]: 0 accept; bit0 challenge mismatch; bit1 ||z|| too large"""
    K, Lv = _kl(level)
    B = z.numel() // (Lv * N)
    verdict = torch.empty((B,), dtype=torch.int32, device=z.device)
    _lib.check(_lib.load().dil_verify_dev(_dev(verdict, torch.int32), _dev(A, torch.int32), _dev(ctilde, torch.uint8),
                                          _dev(z, torch.int32), _dev(t1, torch.int32), _dev(h, torch.uint8),
                                          _dev(mu, torch.uint8), level, B, int(shared_pk), _stream()), "dil_verify_dev")
    return verdict


def sign_attempt(A, mu, rhoprime, kappa, s1hat, s2hat, t0hat, level, shared_key=False):
    """one rejection-loop attempt for every item: returns (ctilde, z, h, flags)"""
    K, Lv = _kl(level)
    B = mu.shape[0]
    ct = torch.empty((B, 32), dtype=torch.uint8, device=mu.device)
    z = torch.empty((B, Lv, N), dtype=torch.int32, device=mu.device)
    h = torch.empty((B, K, N), dtype=torch.uint8, device=mu.device)
    fl = torch.empty((B,), dtype=torch.int32, device=mu.device)
    _lib.check(_lib.load().dil_sign_attempt_dev(_dev(ct, torch.uint8), _dev(z, torch.int32), _dev(h, torch.uint8),
                                                _dev(fl, torch.int32), _dev(A, torch.int32), _dev(mu, torch.uint8),
                                                _dev(rhoprime, torch.uint8), _dev(kappa, torch.int32),
                                                _dev(s1hat, torch.int32), _dev(s2hat, torch.int32), _dev(t0hat, torch.int32),
                                                level, B, int(shared_key), _stream()), "dil_sign_attempt_dev")
    return ct, z, h, fl


# ---- rows N2 / N4: wire-format codecs, ExpandS, keygen, wire-format verify --------------------------
CODEC_T1, CODEC_T0, CODEC_S1, CODEC_S2, CODEC_Z = 0, 1, 2, 3, 4


def _codec_polys(kind, level):
    K, Lv = _kl(level)
    return Lv if kind in (CODEC_S1, CODEC_Z) else K


def pk_bytes(level):
    return int(_lib.load().dil_pk_bytes(level))


def sk_bytes(level):
    return int(_lib.load().dil_sk_bytes(level))


def sig_bytes(level):
    return int(_lib.load().dil_sig_bytes(level))


def unpack(buf, kind, level, offset=0):
    """buf uint8 [B, stride] -> int32 [B, polys, 256] canonical; the field starts at byte `offset` of each row"""
    B, stride = buf.shape
    out = torch.empty((B, _codec_polys(kind, level), N), dtype=torch.int32, device=buf.device)
    _lib.check(_lib.load().dil_unpack_dev(_dev(out, torch.int32), _dev(buf, torch.uint8), stride, offset, kind, level, B, _stream()),
               "dil_unpack_dev")
    return out


def pack(polys, buf, kind, level, offset=0):
    """int32 [B, polys, 256] -> packed field written at byte `offset` of each row of buf uint8 [B, stride]"""
    B, stride = buf.shape
    _lib.check(_lib.load().dil_pack_dev(_dev(buf, torch.uint8), stride, offset, _dev(polys, torch.int32), kind, level, B, _stream()),
               "dil_pack_dev")
    return buf


def hint_unpack(buf, level, offset=0):
    """-> (h uint8 [B,K,256], bad int32 [B])"""
    K, _ = _kl(level)
    B, stride = buf.shape
    h = torch.empty((B, K, N), dtype=torch.uint8, device=buf.device)
    bad = torch.empty((B,), dtype=torch.int32, device=buf.device)
    _lib.check(_lib.load().dil_hint_unpack_dev(_dev(h, torch.uint8), _dev(bad, torch.int32), _dev(buf, torch.uint8), stride, offset,
                                               level, B, _stream()), "dil_hint_unpack_dev")
    return h, bad


def hint_pack(h, buf, level, offset=0):
    B, stride = buf.shape
    _lib.check(_lib.load().dil_hint_pack_dev(_dev(buf, torch.uint8), stride, offset, _dev(h, torch.uint8), level, B, _stream()),
               "dil_hint_pack_dev")
    return buf


def expand_s(rhoprime, level):
    """(s1 [B,L,256], s2 [B,K,256]) canonical from rho' uint8 [B, >=64] (row stride = rhoprime.shape[1])"""
    K, Lv = _kl(level)
    B, stride = rhoprime.shape
    s1 = torch.empty((B, Lv, N), dtype=torch.int32, device=rhoprime.device)
    s2 = torch.empty((B, K, N), dtype=torch.int32, device=rhoprime.device)
    _lib.check(_lib.load().dil_expand_s_dev(_dev(s1, torch.int32), _dev(s2, torch.int32), _dev(rhoprime, torch.uint8), stride,
                                            level, B, _stream()), "dil_expand_s_dev")
    return s1, s2


def keygen(seed, level):
    """seed uint8 [B,32] -> (pk uint8 [B,pk_bytes], sk uint8 [B,sk_bytes])"""
    B = seed.shape[0]
    pk = torch.empty((B, pk_bytes(level)), dtype=torch.uint8, device=seed.device)
    sk = torch.empty((B, sk_bytes(level)), dtype=torch.uint8, device=seed.device)
    _lib.check(_lib.load().dil_keygen_dev(_dev(pk, torch.uint8), _dev(sk, torch.uint8), _dev(seed, torch.uint8), level, B, _stream()),
               "dil_keygen_dev")
    return pk, sk


def verify_sig(pk, sig, mu, level, shared_pk=False):
    """wire-format verify: pk uint8 [B or 1, pk_bytes], sig uint8 [B, sig_bytes], mu uint8 [B,64] -> verdict int32 [B]"""
    B = sig.shape[0]
    verdict = torch.empty((B,), dtype=torch.int32, device=sig.device)
    _lib.check(_lib.load().dil_verify_sig_dev(_dev(verdict, torch.int32), _dev(pk, torch.uint8), _dev(sig, torch.uint8),
                                              _dev(mu, torch.uint8), level, B, int(shared_pk), _stream()), "dil_verify_sig_dev")
    return verdict


def verify_sig_expanded(A, pk, sig, mu, level, shared_pk=False):
    """verify_sig with the keys' matrix A = expand_a(rho) expanded once by the caller and kept across calls"""
    B = sig.shape[0]
    verdict = torch.empty((B,), dtype=torch.int32, device=sig.device)
    _lib.check(_lib.load().dil_verify_sig_expanded_dev(_dev(verdict, torch.int32), _dev(A, torch.int32), _dev(pk, torch.uint8),
                                                       _dev(sig, torch.uint8), _dev(mu, torch.uint8), level, B, int(shared_pk),
                                                       _stream()), "dil_verify_sig_expanded_dev")
    return verdict


def expand_t1(pk, level):
    """t1^ = NTT(t1 2^13) of every key, int32 [B,K,256] canonical -- kept beside expand_a(rho) by a caller that verifies under the same keys again"""
    K, _ = _kl(level)
    B = pk.shape[0]
    out = torch.empty((B, K, N), dtype=torch.int32, device=pk.device)
    _lib.check(_lib.load().dil_expand_t1_dev(_dev(out, torch.int32), _dev(pk, torch.uint8), level, B, _stream()), "dil_expand_t1_dev")
    return out


def verify_sig_expanded2(A, t1hat, pk, sig, mu, level, shared_pk=False):
    """verify_sig with A = expand_a(rho) AND t1^ = expand_t1(pk) of the keys kept by the caller"""
    B = sig.shape[0]
    verdict = torch.empty((B,), dtype=torch.int32, device=sig.device)
    t1p = None if (t1hat is None and shared_pk) else _dev(t1hat, torch.int32)       # one key for the batch: t1^ is not read (include/dil256.h)
    _lib.check(_lib.load().dil_verify_sig_expanded2_dev(_dev(verdict, torch.int32), _dev(A, torch.int32), t1p,
                                                        _dev(pk, torch.uint8), _dev(sig, torch.uint8), _dev(mu, torch.uint8), level, B,
                                                        int(shared_pk), _stream()), "dil_verify_sig_expanded2_dev")
    return verdict


def verify_wire_core(A, pk, sig, level, shared_pk=False):
    """the fused wire-format verify kernel: (w1 packed uint8 [B, K*128|192], verdict int32 [B] with bits 2 | 4)"""
    K, _ = _kl(level)
    B = sig.shape[0]
    w1p = torch.empty((B, K * (192 if level == 2 else 128)), dtype=torch.uint8, device=sig.device)
    verdict = torch.empty((B,), dtype=torch.int32, device=sig.device)
    _lib.check(_lib.load().dil_verify_wire_core_dev(_dev(w1p, torch.uint8), _dev(verdict, torch.int32), None if A is None else _dev(A, torch.int32),
                                                    _dev(pk, torch.uint8), _dev(sig, torch.uint8), level, B, int(shared_pk),
                                                    _stream()), "dil_verify_wire_core_dev")
    return w1p, verdict


def sign(sk, mu, level, shared_sk=False, max_attempts=512):
    """wire-format deterministic signing: sk uint8 [B or 1, sk_bytes], mu uint8 [B,64] -> (sig uint8 [B,sig_bytes], attempts int32 [B])"""
    B = mu.shape[0]
    sig = torch.empty((B, sig_bytes(level)), dtype=torch.uint8, device=mu.device)
    att = torch.empty((B,), dtype=torch.int32, device=mu.device)
    _lib.check(_lib.load().dil_sign_dev(_dev(sig, torch.uint8), _dev(att, torch.int32), _dev(sk, torch.uint8), _dev(mu, torch.uint8),
                                        level, B, int(shared_sk), max_attempts, _stream()), "dil_sign_dev")
    return sig, att


# ---- messages in, not digests: mu = SHAKE256(tr || M) on the device ---------------------------------------------------
def pack_messages(msgs, device="cuda"):
    """list of bytes -> (blob uint8 [sum len], offsets int64 [B], lengths int32 [B]) device tensors (ragged batch)"""
    lens = np.array([len(m) for m in msgs], dtype=np.int32)
    offs = np.zeros(len(msgs), dtype=np.int64)
    offs[1:] = np.cumsum(lens[:-1], dtype=np.int64)
    blob = np.frombuffer(b"".join(msgs), dtype=np.uint8).copy()          # may be empty: the library takes (NULL, 0)
    return torch.from_numpy(blob).to(device), torch.from_numpy(offs).to(device), torch.from_numpy(lens).to(device)


def _blob(blob):
    return (C.c_void_p(blob.data_ptr()) if blob.numel() else None), blob.numel()


def mu(tr, blob, offsets, lengths, bad=None):
    """mu uint8 [B,64] = SHAKE256(tr_i || M_i); tr uint8 [B,32] or [1,32] (one tr for the batch); bad (optional int32 [B]):
    1 where (offset, length) leaves the blob (that item is hashed as an empty message)"""
    B = lengths.shape[0]
    out = torch.empty((B, 64), dtype=torch.uint8, device=lengths.device)
    stride = 0 if tr.shape[0] == 1 and B > 1 else tr.stride(0)
    bp, bn = _blob(blob)
    _lib.check(_lib.load().dil_mu_dev(_dev(out, torch.uint8), C.c_void_p(tr.data_ptr()), stride, bp, bn,
                                      _dev(offsets, torch.int64), _dev(lengths, torch.int32),
                                      None if bad is None else _dev(bad, torch.int32), B, _stream()), "dil_mu_dev")
    return out


def sign_msg(sk, blob, offsets, lengths, level, shared_sk=False, max_attempts=512):
    """deterministic signing of ragged messages: (sig uint8 [B,sig_bytes], attempts int32 [B])"""
    B = lengths.shape[0]
    sig = torch.empty((B, sig_bytes(level)), dtype=torch.uint8, device=sk.device)
    att = torch.empty((B,), dtype=torch.int32, device=sk.device)
    bp, bn = _blob(blob)
    _lib.check(_lib.load().dil_sign_msg_dev(_dev(sig, torch.uint8), _dev(att, torch.int32), _dev(sk, torch.uint8), bp, bn,
                                            _dev(offsets, torch.int64), _dev(lengths, torch.int32), level, B, int(shared_sk),
                                            max_attempts, _stream()), "dil_sign_msg_dev")
    return sig, att


def verify_msg(pk, sig, blob, offsets, lengths, level, shared_pk=False):
    """verification of (pk, M, sig): verdict int32 [B], 0 = accept"""
    B = sig.shape[0]
    verdict = torch.empty((B,), dtype=torch.int32, device=sig.device)
    bp, bn = _blob(blob)
    _lib.check(_lib.load().dil_verify_msg_dev(_dev(verdict, torch.int32), _dev(pk, torch.uint8), _dev(sig, torch.uint8),
                                              bp, bn, _dev(offsets, torch.int64), _dev(lengths, torch.int32),
                                              level, B, int(shared_pk), _stream()), "dil_verify_msg_dev")
    return verdict


# ---- host-buffer forms (numpy uint8 arrays in, numpy out) ---------------------------------------------
def _np8(a, cols=None):
    if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags.c_contiguous and a.ndim == 2):
        raise ValueError("expected a C-contiguous 2-D numpy uint8 array")
    if cols is not None and a.shape[1] != cols:
        raise ValueError(f"expected rows of {cols} bytes, got {a.shape[1]}")
    return C.c_void_p(a.ctypes.data)


def keygen_host(seed, level):
    B = seed.shape[0]
    pk = np.empty((B, pk_bytes(level)), dtype=np.uint8)
    sk = np.empty((B, sk_bytes(level)), dtype=np.uint8)
    _lib.check(_lib.load().dil_keygen_host(_np8(pk), _np8(sk), _np8(seed, 32), level, B), "dil_keygen_host")
    return pk, sk


def sign_host(sk, mu, level, shared_sk=False, max_attempts=512):
    B = mu.shape[0]
    sig = np.empty((B, sig_bytes(level)), dtype=np.uint8)
    att = np.empty((B,), dtype=np.int32)
    _lib.check(_lib.load().dil_sign_host(_np8(sig), C.c_void_p(att.ctypes.data), _np8(sk, sk_bytes(level)), _np8(mu, 64), level, B,
                                         int(shared_sk), max_attempts), "dil_sign_host")
    return sig, att


def verify_sig_host(pk, sig, mu, level, shared_pk=False):
    B = sig.shape[0]
    verdict = np.empty((B,), dtype=np.int32)
    _lib.check(_lib.load().dil_verify_sig_host(C.c_void_p(verdict.ctypes.data), _np8(pk, pk_bytes(level)), _np8(sig, sig_bytes(level)),
                                               _np8(mu, 64), level, B, int(shared_pk)), "dil_verify_sig_host")
    return verdict
