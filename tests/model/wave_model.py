"""Executable spec of the HIP kernels' wave-level dataflow (numpy, 64 lanes x 4 registers).

TEST INFRASTRUCTURE.  Mirrors dilithium_amd/csrc/ntt_core.hpp step for step -- same lane
layout, same cross-lane exchanges, same lazy-reduction schedule, same twiddle tables --
so that layout / twiddle-index / overflow-bound mistakes are caught on CPU (no GPU in the
dev container).  Every multiply asserts its 24-bit operand contract; every add asserts
no 32-bit overflow.
"""
import numpy as np

Q = 8380417
N = 256
LANES = 64
F256 = 8347681  # 256^-1 mod q (ref_ntt.cpp:64)


def brv8(x):
    return int(f"{x:08b}"[::-1], 2)


ZETA = [0] + [pow(1753, brv8(k), Q) for k in range(1, N)]  # canonical zetas (== zetas.txt)


def shoup(w):
    return (w << 24) // Q


# ---- arithmetic primitives (uint32 semantics, checked) -------------------------------
def mul24(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    assert (a < (1 << 24)).all() and (b < (1 << 24)).all(), "u24 operand out of range"
    return a * b  # full 48-bit product (caller takes what it needs)


def red(x):
    """x - (x>>23)*q : any uint32 -> [0, 2^23 + 2^22)"""
    x = np.asarray(x, dtype=np.uint64)
    assert (x < (1 << 32)).all(), "32-bit overflow"
    r = x - (x >> np.uint64(23)) * np.uint64(Q)
    assert (r < (1 << 23) + (1 << 22)).all()
    return r


def shoup_mul(y, w, wp):
    """y < 2^24, w < q, wp = floor(w 2^24 / q) -> y*w mod q in [0, 2q)"""
    p = mul24(y, wp)
    qe = p >> np.uint64(24)
    r = (mul24(y, w) - mul24(qe, Q))
    assert (r < 2 * Q).all()
    return r


def canon_final(x):
    r = red(x)
    return np.where(r >= Q, r - Q, r)


# ---- cross-lane exchanges: 4x4 transpose between register index and a lane bit-pair ---
def xchg(r, shift):
    """r: [4][64].  Transpose reg index (2 bits) with lane bits [shift+1:shift]."""
    out = np.empty_like(r)
    lane = np.arange(LANES)
    grp = (lane >> shift) & 3
    for m in range(4):            # new register index m
        for c in range(4):        # lanes whose group == c receive old register c from lane with group m
            sel = grp == c
            src_lane = (lane & ~(3 << shift)) | (m << shift)
            out[m][sel] = r[c][src_lane[sel]]
    return out


# ---- twiddle tables, exactly as the host library builds them ---------------------------
def fwd_table():
    """[4 passes][6 = (wa, wa', wb0, wb0', wb1, wb1')][64 lanes]"""
    t = np.zeros((4, 6, LANES), dtype=np.uint64)
    for p in range(4):
        for lane in range(LANES):
            k1 = (1 << (2 * p)) + (lane >> (6 - 2 * p))
            ws = [ZETA[k1], ZETA[2 * k1], ZETA[2 * k1 + 1]]
            for i, w in enumerate(ws):
                t[p, 2 * i, lane] = w
                t[p, 2 * i + 1, lane] = shoup(w)
    return t


def inv_table():
    """[4 passes][8 = (wa0, wa0', wa1, wa1', wb, wb', f, f')][64 lanes]; the last pass's wb is
    pre-multiplied by f = 256^-1 and f itself rides along (the 1/256 of ref_ntt.cpp:83-86)."""
    t = np.zeros((4, 8, LANES), dtype=np.uint64)
    for p in range(4):
        for lane in range(LANES):
            blk = lane >> (2 * p) if p < 3 else 0
            base = N >> (2 * p)
            ka = base - 1 - 2 * blk
            kb = (base >> 1) - 1 - blk
            ws = [(Q - ZETA[ka]) % Q, (Q - ZETA[ka - 1]) % Q, (Q - ZETA[kb]) % Q]
            if p == 3:
                ws[2] = ws[2] * F256 % Q
            ws.append(F256)
            for i, w in enumerate(ws):
                t[p, 2 * i, lane] = w
                t[p, 2 * i + 1, lane] = shoup(w)
    return t


FWD = fwd_table()
INV = inv_table()


# ---- butterflies ---------------------------------------------------------------------------
def ct(x, y, w, wp):
    t = shoup_mul(red(y), w, wp)
    xn = x + t
    yn = x + np.uint64(2 * Q) - t
    assert (xn < (1 << 32)).all() and (yn < (1 << 32)).all()
    return xn, yn


def gs(x, y, w, wp, by):
    """by = static bound (in units of q) on y"""
    s = x + y
    d = x + np.uint64(by * Q) - y
    assert (s < (1 << 32)).all() and (d < (1 << 32)).all() and (y <= by * Q).all()
    return s, shoup_mul(red(d), w, wp)


# ---- forward NTT: natural in (any int32 in [-q, 2^31)) -> reference order, canonical ----------
def ntt_wave(a, exchanges_out=None):
    """a: int array[256].  Returns (r [4][64] with lane j holding out[4j..4j+3], out[256])."""
    a = np.asarray(a, dtype=np.int64)
    lane = np.arange(LANES)
    r = np.stack([(a[lane + 64 * m] + Q).astype(np.uint64) for m in range(4)])  # strided load, +q
    assert (r < (1 << 32)).all()
    for p in range(4):
        wa, wap, wb0, wb0p, wb1, wb1p = FWD[p]
        r0, r2 = ct(r[0], r[2], wa, wap)
        r1, r3 = ct(r[1], r[3], wa, wap)
        r0, r1 = ct(r0, r1, wb0, wb0p)
        r2, r3 = ct(r2, r3, wb1, wb1p)
        r = np.stack([r0, r1, r2, r3])
        if p < 3:
            r = xchg(r, 4 - 2 * p)
    r = np.stack([canon_final(x) for x in r])
    out = np.empty(N, dtype=np.int64)
    for m in range(4):
        out[4 * lane + m] = r[m]
    return r, out


# ---- inverse NTT: reference order in (lane j holds a[4j..4j+3]) -> natural, canonical ----------
def invntt_wave(a):
    a = np.asarray(a, dtype=np.int64)
    lane = np.arange(LANES)
    r = np.stack([(a[4 * lane + m] + Q).astype(np.uint64) for m in range(4)])
    for p in range(4):
        wa0, wa0p, wa1, wa1p, wb, wbp, f, fp = INV[p]
        # every register enters a pass below 2q (loads: x+q; later passes: see the reds below)
        s01, m01 = gs(r[0], r[1], wa0, wa0p, 2)
        s23, m23 = gs(r[2], r[3], wa1, wa1p, 2)       # s: 4q, m: 2q
        if p < 3:
            s02, m02 = gs(s01, s23, wb, wbp, 4)       # 8q, 2q
            s13, m13 = gs(m01, m23, wb, wbp, 2)       # 4q, 2q
            # the sums carry 8q / 4q: pull them under 2q BEFORE the exchange mixes registers
            r = np.stack([red(s02), red(s13), m02, m13])
            r = xchg(r, 2 * p)
        else:
            s02, m02 = gs(s01, s23, wb, wbp, 4)
            s13, m13 = gs(m01, m23, wb, wbp, 2)
            s02 = shoup_mul(red(s02), f, fp)
            s13 = shoup_mul(red(s13), f, fp)
            r = np.stack([s02, s13, m02, m13])
    r = np.stack([np.where(x >= Q, x - Q, x) for x in r])   # [0,2q) -> [0,q)
    out = np.empty(N, dtype=np.int64)
    for m in range(4):
        out[lane + 64 * m] = r[m]
    return r, out
