"""Executable spec of the HIP kernels' wave-level dataflow (numpy, 64 lanes x 4 registers).

TEST INFRASTRUCTURE.  Mirrors dilithium_amd/csrc/{modarith,ntt_core}.hpp step for step --
same lane layout, same cross-lane exchanges, same signed-Montgomery arithmetic (R = 2^32),
same twiddle tables -- so that layout / twiddle-index / overflow mistakes are caught on CPU
(the dev container has no GPU).  Every intermediate asserts that it fits in int32 and every
Montgomery product asserts its |a*b| < 2^31 * q contract.
"""
import numpy as np

Q = 8380417
N = 256
LANES = 64
F256 = 8347681            # 256^-1 mod q (ref_ntt.cpp:64)
R = 1 << 32
QINV = pow(Q, -1, R)      # 58728449
assert QINV == 58728449


def brv8(x):
    return int(f"{x:08b}"[::-1], 2)


ZETA = [0] + [pow(1753, brv8(k), Q) for k in range(1, N)]   # canonical zetas (== zetas.txt)


def centered(x):
    x %= Q
    return x - Q if x > (Q - 1) // 2 else x


def mont_const(w):
    """(w~, wq): w~ = centred(w * 2^32 mod q),  wq = w~ * q^-1 mod 2^32 (as a 32-bit pattern)"""
    wt = centered(w * R % Q)
    return wt, (wt * QINV) % R


def i32(x):
    x = np.asarray(x, dtype=np.int64)
    assert (x >= -(1 << 31)).all() and (x < (1 << 31)).all(), "int32 overflow"
    return x


def wrap32(x):
    """two's-complement wrap of an int64 array to signed 32 bit"""
    x = np.asarray(x, dtype=np.int64) & 0xFFFFFFFF
    return np.where(x >= (1 << 31), x - (1 << 32), x)


def mulhi(a, b):
    return (np.asarray(a, dtype=np.int64) * np.asarray(b, dtype=np.int64)) >> 32   # floor, like v_mul_hi_i32


def mont_tw(y, wt, wq):
    """y * w (true product, w given in Montgomery form): 3 multiplies + 1 subtract.  |result| < q"""
    y = i32(y)
    wt = np.asarray(wt, dtype=np.int64)
    assert (np.abs(y * wt) < (1 << 31) * Q).all()
    m = wrap32(y * np.asarray(wq, dtype=np.int64))
    t = mulhi(y, wt) - mulhi(m, Q)
    assert (np.abs(t) < Q).all()
    return t


def mont_red64(p):
    """p (int64, |p| < 2^31 q) -> p * 2^-32 mod q, |result| < q"""
    p = np.asarray(p, dtype=np.int64)
    assert (np.abs(p) < (1 << 31) * Q).all()
    lo = wrap32(p)
    m = wrap32(lo * QINV)
    t = (p >> 32) - mulhi(m, Q)
    assert (np.abs(t) < Q).all()
    return t


def canon_any(x):
    """any int32 -> [0, q)"""
    x = i32(x)
    k = (x + (1 << 22)) >> 23
    r = x - k * Q
    assert (np.abs(r) < Q).all()
    return np.where(r < 0, r + Q, r)


def canon_small(t):
    """(-q, q) -> [0, q)"""
    assert (np.abs(t) < Q).all()
    return np.where(t < 0, t + Q, t)


# ---- cross-lane exchanges: 4x4 transpose between register index and a lane bit-pair ---
def xchg(r, shift):
    out = np.empty_like(r)
    lane = np.arange(LANES)
    grp = (lane >> shift) & 3
    for m in range(4):
        for c in range(4):
            sel = grp == c
            src_lane = (lane & ~(3 << shift)) | (m << shift)
            out[m][sel] = r[c][src_lane[sel]]
    return out


# ---- twiddle tables, exactly as the host library builds them: [pass][lane][8] ------------
def fwd_table():
    t = np.zeros((4, LANES, 8), dtype=np.int64)
    for p in range(4):
        for lane in range(LANES):
            k1 = (1 << (2 * p)) + (lane >> (6 - 2 * p))
            for i, w in enumerate([ZETA[k1], ZETA[2 * k1], ZETA[2 * k1 + 1]]):
                t[p, lane, 2 * i], t[p, lane, 2 * i + 1] = mont_const(w)
    return t


def inv_table(pipeline):
    """last pass: wb *= f and slots 6,7 = f  (f = 256^-1; x 2^32 more in the pipeline flavour,
    which cancels the 2^-32 left by the Montgomery reduction of the pointwise products)"""
    t = np.zeros((4, LANES, 8), dtype=np.int64)
    f = F256 * (R % Q) % Q if pipeline else F256
    for p in range(4):
        for lane in range(LANES):
            blk = lane >> (2 * p) if p < 3 else 0
            base = N >> (2 * p)
            ka = base - 1 - 2 * blk
            kb = (base >> 1) - 1 - blk
            ws = [(Q - ZETA[ka]) % Q, (Q - ZETA[ka - 1]) % Q, (Q - ZETA[kb]) % Q, f]
            if p == 3:
                ws[2] = ws[2] * f % Q
            for i, w in enumerate(ws):
                t[p, lane, 2 * i], t[p, lane, 2 * i + 1] = mont_const(w)
    return t


FWD = fwd_table()
INV = inv_table(False)
INV_PIPE = inv_table(True)


def table_u32(t):
    return (t & 0xFFFFFFFF).astype(np.uint32)


# ---- butterflies ---------------------------------------------------------------------------
def ct(x, y, wt, wq):
    t = mont_tw(y, wt, wq)
    return i32(x + t), i32(x - t)


def gs(x, y, wt, wq):
    return i32(x + y), mont_tw(i32(x - y), wt, wq)


def fwd_core(r):
    for p in range(4):
        T = FWD[p].T
        r0, r2 = ct(r[0], r[2], T[0], T[1])
        r1, r3 = ct(r[1], r[3], T[0], T[1])
        r0, r1 = ct(r0, r1, T[2], T[3])
        r2, r3 = ct(r2, r3, T[4], T[5])
        r = np.stack([r0, r1, r2, r3])
        if p < 3:
            r = xchg(r, 4 - 2 * p)
    return r


def inv_core(r, table):
    for p in range(4):
        T = table[p].T
        r0, r1 = gs(r[0], r[1], T[0], T[1])
        r2, r3 = gs(r[2], r[3], T[2], T[3])
        r0, r2 = gs(r0, r2, T[4], T[5])
        r1, r3 = gs(r1, r3, T[4], T[5])
        if p < 3:
            r = xchg(np.stack([r0, r1, r2, r3]), 2 * p)
        else:
            r = np.stack([mont_tw(r0, T[6], T[7]), mont_tw(r1, T[6], T[7]), r2, r3])
    return r


def ntt_wave(a):
    """forward NTT: natural in (int32) -> reference order, canonical"""
    a = i32(a)
    lane = np.arange(LANES)
    r = fwd_core(np.stack([a[lane + 64 * m] for m in range(4)]))
    r = np.stack([canon_any(x) for x in r])
    out = np.empty(N, dtype=np.int64)
    for m in range(4):
        out[4 * lane + m] = r[m]
    return r, out


def invntt_wave(a, table=None):
    """inverse NTT: reference order in, |a| < q -> natural order, canonical"""
    a = i32(a)
    assert (np.abs(a) < Q).all()
    lane = np.arange(LANES)
    r = inv_core(np.stack([a[4 * lane + m] for m in range(4)]), INV if table is None else table)
    r = np.stack([canon_small(x) for x in r])
    out = np.empty(N, dtype=np.int64)
    for m in range(4):
        out[lane + 64 * m] = r[m]
    return r, out


def matvec_wave(A, y):
    """w[k] = INTT(sum_l A[k][l] o NTT(y[l])) the way the fused kernels do it: lazy NTT outputs,
    64-bit multiply-accumulate, one Montgomery reduction, pipeline-flavour inverse table"""
    K, L = A.shape[:2]
    lane = np.arange(LANES)
    yh = []
    for l in range(L):
        r = fwd_core(np.stack([i32(y[l])[lane + 64 * m] for m in range(4)]))     # lazy, |x| <= 7q
        yh.append(r)
    out = np.empty((K, N), dtype=np.int64)
    for k in range(K):
        acc = np.zeros((4, LANES), dtype=np.int64)
        for l in range(L):
            Arow = np.stack([A[k, l][4 * lane + m] for m in range(4)]).astype(np.int64)
            acc += Arow * yh[l]
        r = np.stack([mont_red64(x) for x in acc])
        r = inv_core(r, INV_PIPE)
        r = np.stack([canon_small(x) for x in r])
        for m in range(4):
            out[k, lane + 64 * m] = r[m]
    return out
