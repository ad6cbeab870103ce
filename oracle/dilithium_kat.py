"""KAT harness: CRYSTALS-Dilithium round-3 v3.1 keygen / sign / verify in Python.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Written from the Dilithium round-3
specification and SURVEY.md App. A (conventions the reference's KAT files obey), with
hashlib SHAKE for every hash and the C oracle (oracle/dil_oracle.c) for polynomial
arithmetic.  The RTL this mirrors: combined_top.v keygen :754-1079, verify :1080-1534,
sign :1535-2232; samplers gen_a_ext.v / gen_s.v / expandmask_ext.v / gen_c.v; codecs
decoder.v / encoder.v.  The polynomial "engine" is pluggable so the same driver can push
the KATs through the HIP kernels (tests/) or through the oracle.

KAT file layout (reference KAT/*.txt, one hex line per vector): rho,tr,k,c,z(=keygen seed)
32 B; s1 L*(96|128) B; s2 K*(96|128); t0 K*416; t1 K*320; zs (signature z) L*(576|640);
h omega+K; m padded to 3300 B; mlen 2 B big-endian (33*(i+1)).
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass

import numpy as np

Q = 8380417
N = 256
D = 13


@dataclass(frozen=True)
class Params:
    level: int
    K: int
    L: int
    eta: int
    tau: int
    gamma1: int
    gamma2: int
    omega: int
    beta: int

    @property
    def z_bits(self):
        return 18 if self.gamma1 == (1 << 17) else 20

    @property
    def w1_bits(self):
        return 6 if self.gamma2 == (Q - 1) // 88 else 4

    @property
    def eta_bits(self):
        return 3 if self.eta == 2 else 4


PARAMS = {
    2: Params(2, 4, 4, 2, 39, 1 << 17, (Q - 1) // 88, 80, 78),
    3: Params(3, 6, 5, 4, 49, 1 << 19, (Q - 1) // 32, 55, 196),
    5: Params(5, 8, 7, 2, 60, 1 << 19, (Q - 1) // 32, 75, 120),
}


def shake256(data: bytes, n: int) -> bytes:
    return hashlib.shake_256(data).digest(n)


def shake128(data: bytes, n: int) -> bytes:
    return hashlib.shake_128(data).digest(n)


# ---------------------------------------------------------------- bit packing
def pack_bits(vals: np.ndarray, bits: int) -> bytes:
    """little-endian bit stream of `bits`-bit unsigned values"""
    v = np.asarray(vals, dtype=np.uint64).reshape(-1)
    b = ((v[:, None] >> np.arange(bits, dtype=np.uint64)) & np.uint64(1)).astype(np.uint8).reshape(-1)
    return np.packbits(b, bitorder="little").tobytes()


def unpack_bits(data: bytes, bits: int, count: int) -> np.ndarray:
    b = np.unpackbits(np.frombuffer(data, dtype=np.uint8), bitorder="little")[: bits * count]
    w = (np.uint64(1) << np.arange(bits, dtype=np.uint64))
    return (b.reshape(count, bits).astype(np.uint64) * w).sum(axis=1).astype(np.int64)


def canon(a):
    return np.mod(np.asarray(a, dtype=np.int64), Q).astype(np.int32)


def centered(a):
    a = np.mod(np.asarray(a, dtype=np.int64), Q)
    return np.where(a > (Q - 1) // 2, a - Q, a)


def unpack_t1(p, data):  # 10 b
    return unpack_bits(data, 10, p.K * N).reshape(p.K, N).astype(np.int32)


def pack_t1(p, t1):
    return pack_bits(t1, 10)


def unpack_t0(p, data):  # 13 b as 2^12 - t0
    return ((1 << (D - 1)) - unpack_bits(data, 13, p.K * N)).reshape(p.K, N)


def pack_t0(p, t0):
    return pack_bits((1 << (D - 1)) - centered(t0), 13)


def unpack_eta(p, data, rows):
    return (p.eta - unpack_bits(data, p.eta_bits, rows * N)).reshape(rows, N)


def pack_eta(p, s):
    return pack_bits(p.eta - centered(s), p.eta_bits)


def unpack_z(p, data):
    return (p.gamma1 - unpack_bits(data, p.z_bits, p.L * N)).reshape(p.L, N)


def pack_z(p, z):
    return pack_bits(p.gamma1 - centered(z), p.z_bits)


def pack_w1(p, w1):
    return pack_bits(w1, p.w1_bits)


def unpack_hint(p, data):
    """omega position bytes then K cumulative counts -> h [K][256] 0/1, or None if malformed"""
    h = np.zeros((p.K, N), dtype=np.uint8)
    k = 0
    for i in range(p.K):
        end = data[p.omega + i]
        if end < k or end > p.omega:
            return None
        for j in range(k, end):
            if j > k and data[j] <= data[j - 1]:
                return None
            h[i, data[j]] = 1
        k = end
    if any(data[j] != 0 for j in range(k, p.omega)):
        return None
    return h


def pack_hint(p, h):
    out = bytearray(p.omega + p.K)
    k = 0
    for i in range(p.K):
        for j in np.nonzero(h[i])[0]:
            out[k] = int(j)
            k += 1
        out[p.omega + i] = k
    return bytes(out)


# ------------------------------------------------------------------- samplers
def expand_a_poly(rho: bytes, i: int, j: int) -> np.ndarray:
    """A[i][j]: SHAKE128(rho || byte j || byte i), 3-byte LE, 23-bit mask, accept < q
    (sampler_a_ext.v:129 nonce (i<<8)|j, rejection_a.v:67-73)"""
    nbytes = 840
    while True:
        buf = np.frombuffer(shake128(rho + bytes([j, i]), nbytes), dtype=np.uint8).reshape(-1, 3).astype(np.int64)
        v = (buf[:, 0] | (buf[:, 1] << 8) | (buf[:, 2] << 16)) & 0x7FFFFF
        v = v[v < Q]
        if v.size >= N:
            return v[:N].astype(np.int32)
        nbytes += 168 * 3


def expand_a(p, rho: bytes) -> np.ndarray:
    return np.stack([np.stack([expand_a_poly(rho, i, j) for j in range(p.L)]) for i in range(p.K)])


def expand_s_poly(p, rhop: bytes, nonce: int) -> np.ndarray:
    """gen_s.v / rejection_s.v: SHAKE256(rho' || LE16 nonce), nibble rejection"""
    nbytes = 272
    while True:
        buf = np.frombuffer(shake256(rhop + nonce.to_bytes(2, "little"), nbytes), dtype=np.uint8)
        nib = np.stack([buf & 15, buf >> 4], axis=1).reshape(-1).astype(np.int64)
        if p.eta == 2:
            nib = nib[nib < 15]
            v = 2 - (nib - (205 * nib >> 10) * 5)
        else:
            nib = nib[nib < 9]
            v = 4 - nib
        if v.size >= N:
            return v[:N]
        nbytes += 136


def expand_mask_poly(p, rhop: bytes, nonce: int) -> np.ndarray:
    """expandmask_ext.v:98 / sampler_y_ext.v: y = gamma1 - unpack(SHAKE256(rho' || LE16 nonce))"""
    nb = N * p.z_bits // 8
    return p.gamma1 - unpack_bits(shake256(rhop + nonce.to_bytes(2, "little"), nb), p.z_bits, N)


def sample_in_ball(p, ctilde: bytes) -> np.ndarray:
    """gen_c.v:163-196,318-339: SHAKE256(c~): 8 sign bytes, then rejection b <= i"""
    buf = shake256(ctilde, 8 + 136 * 8)
    signs = int.from_bytes(buf[:8], "little")
    pos = 8
    c = np.zeros(N, dtype=np.int64)
    for i in range(N - p.tau, N):
        while True:
            b = buf[pos]
            pos += 1
            if b <= i:
                break
        c[i] = c[b]
        c[b] = 1 - 2 * (signs & 1)
        signs >>= 1
    return c


# ------------------------------------------------------------- poly engines
class OracleEngine:
    """polynomial back-end = the C oracle.  Same interface as tests' HIP engine."""

    def __init__(self, oracle=None):
        from .oracle import Oracle
        self.o = oracle or Oracle()

    def ntt(self, a):
        return self.o.ntt(canon(a))

    def matvec(self, p, A, y):            # y [n][L][256] canonical -> w [n][K][256]
        return self.o.matvec(p.K, p.L, A, canon(y))

    def verify_core(self, p, A, z, c, t1, h):
        return self.o.verify_core(p.level, A, canon(z), canon(c), t1, h)

    def sign_phase1(self, p, A, y):
        return self.o.sign_phase1(p.level, A, canon(y))

    def sign_phase2(self, p, c, y, w0, w1, s1h, s2h, t0h):
        return self.o.sign_phase2(p.level, canon(c), canon(y), w0, w1, s1h, s2h, t0h)


# ------------------------------------------------------------------- schemes
def power2round(a):
    a = np.asarray(a, dtype=np.int64)
    a1 = (a + (1 << (D - 1)) - 1) >> D
    return a1, a - (a1 << D)


def keygen(level, seed: bytes, eng):
    p = PARAMS[level]
    buf = shake256(seed, 128)
    rho, rhop, key = buf[:32], buf[32:96], buf[96:128]
    A = expand_a(p, rho)
    s1 = np.stack([expand_s_poly(p, rhop, i) for i in range(p.L)])
    s2 = np.stack([expand_s_poly(p, rhop, p.L + i) for i in range(p.K)])
    w = eng.matvec(p, A[None], canon(s1)[None])[0].astype(np.int64)
    t = np.mod(w + s2, Q)
    t1, t0 = power2round(t)
    pk_t1 = pack_t1(p, t1)
    tr = shake256(rho + pk_t1, 32)
    return dict(rho=rho, key=key, tr=tr, s1=s1, s2=s2, t1=t1.astype(np.int32), t0=t0, t1_packed=pk_t1)


def verify_batch(level, items, eng):
    """items: list of dict(rho, ctilde, z_packed, t1_packed, h_packed, msg).  Returns list of bool.
    One batched call into the engine's verify core (config 4 of BASELINE.json)."""
    p = PARAMS[level]
    n = len(items)
    ok = [True] * n
    A = np.empty((n, p.K, p.L, N), np.int32)
    z = np.empty((n, p.L, N), np.int32)
    c = np.empty((n, N), np.int32)
    t1 = np.empty((n, p.K, N), np.int32)
    h = np.zeros((n, p.K, N), np.uint8)
    mus = []
    for i, it in enumerate(items):
        A[i] = expand_a(p, it["rho"])
        zz = unpack_z(p, it["z_packed"])
        if np.abs(zz).max() >= p.gamma1 - p.beta:
            ok[i] = False
        z[i] = canon(zz)
        c[i] = canon(sample_in_ball(p, it["ctilde"]))
        t1[i] = unpack_t1(p, it["t1_packed"])
        hh = unpack_hint(p, it["h_packed"])
        if hh is None:
            ok[i] = False
        else:
            h[i] = hh
        tr = shake256(it["rho"] + it["t1_packed"], 32)
        mus.append(shake256(tr + it["msg"], 64))
    w1 = eng.verify_core(p, A, z, c, t1, h)
    for i in range(n):
        if ok[i]:
            ok[i] = shake256(mus[i] + pack_w1(p, w1[i]), 32) == items[i]["ctilde"]
    return ok, w1


def sign_batch(level, items, eng, max_attempts=64):
    """Deterministic signing of a batch (config 5 of BASELINE.json: the sign inner loop).
    items: dict(rho, key, tr, s1_packed, s2_packed, t0_packed, msg).  All signatures run
    their attempt #n together (one phase-1 and one phase-2 engine call per round);
    returns list of (ctilde, z_packed, h_packed, attempts)."""
    p = PARAMS[level]
    n = len(items)
    A = np.stack([expand_a(p, it["rho"]) for it in items])
    s1h = np.stack([eng.ntt(canon(unpack_eta(p, it["s1_packed"], p.L))) for it in items])
    s2h = np.stack([eng.ntt(canon(unpack_eta(p, it["s2_packed"], p.K))) for it in items])
    t0h = np.stack([eng.ntt(canon(unpack_t0(p, it["t0_packed"]))) for it in items])
    mu = [shake256(it["tr"] + it["msg"], 64) for it in items]
    rhop = [shake256(it["key"] + m, 64) for it, m in zip(items, mu)]
    out = [None] * n
    live = list(range(n))
    kappa = 0
    attempts = 0
    while live and attempts < max_attempts:
        attempts += 1
        y = np.stack([np.stack([expand_mask_poly(p, rhop[i], kappa + j) for j in range(p.L)]) for i in live])
        kappa += p.L
        yc = canon(y)
        w1, w0 = eng.sign_phase1(p, A[live], yc)
        ct = [shake256(mu[i] + pack_w1(p, w1[j]), 32) for j, i in enumerate(live)]
        c = np.stack([canon(sample_in_ball(p, x)) for x in ct])
        z, h, flags = eng.sign_phase2(p, c, yc, w0, w1, s1h[live], s2h[live], t0h[live])
        nxt = []
        for j, i in enumerate(live):
            if flags[j] == 0:
                out[i] = (ct[j], pack_z(p, z[j]), pack_hint(p, h[j]), attempts)
            else:
                nxt.append(i)
        live = nxt
    return out


# ------------------------------------------------------------------ KAT files
def load_kat_reference(level, kat_dir="/root/reference/KAT"):
    """parse the reference's KAT text files (dev container only)"""
    def lines(name):
        with open(f"{kat_dir}/{name}_{level}.txt") as f:
            return [bytes.fromhex(x.strip()) for x in f.read().split()]
    mlen = [int.from_bytes(x, "big") for x in lines("mlen")]
    m = lines("m")
    d = dict(seed=lines("z"), rho=lines("rho"), key=lines("k"), tr=lines("tr"), ctilde=lines("c"),
             s1=lines("s1"), s2=lines("s2"), t0=lines("t0"), t1=lines("t1"), z=lines("zs"), h=lines("h"),
             msg=[mm[:l] for mm, l in zip(m, mlen)])
    return d
