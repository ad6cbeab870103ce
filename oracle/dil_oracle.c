/*
 * dil_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the arithmetic of the one hot path this repository
 * accelerates (batched n=256 NTT / INTT / pointwise multiply-accumulate over
 * Z_q, q = 8380417, and the Dilithium mat-vec / verify-core / sign-inner-loop
 * pipelines built from them).  It exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can CHECK the HIP path.  Nothing under
 * dilithium_amd/ may import, link or call it: the product path is HIP only.
 *
 * Every function cites the reference file:line (relative to the GMUCERG/Dilithium
 * tree) whose behaviour it restates.  It is written from the algorithm, not
 * copied: the twiddle table is computed (zeta = 1753) rather than pasted, the
 * hardware model is restated semantically (address translation + transform)
 * instead of FIFO-by-FIFO.
 *
 * Pinning (see oracle/README.md, tests/test_oracle_*.py):
 *   - twiddles  == zetas.txt ROM image (tests/golden/zetas.txt)          [golden]
 *   - ntt/invntt/pointwise/2x2/bram ops == compiled reference (oracle/_ref)
 *     on seeded random + edge polynomials; outputs committed in tests/golden
 *   - verify core / mat-vec / sign loop == the reference's 100 KAT vectors
 *     at levels 2/3/5 through oracle/dilithium_kat.py
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include <time.h>

#define Q 8380417
#define N 256
#define LOGN 8

/* ------------------------------------------------------------------ */
/* H1: twiddles.  consts.cpp:64-97 holds zeta^brv8(k) (zeta = 1753) as    */
/* centred representatives in [-(q-1)/2, (q-1)/2], entry 0 unused (= 0). */
/* zetas.txt holds the same values mod q.  We compute them.              */
/* ------------------------------------------------------------------ */
static int32_t ZETAS[N];
static int zetas_ready = 0;

static uint32_t brv8(uint32_t x)
{
    uint32_t r = 0;
    for (int i = 0; i < 8; i++) r |= ((x >> i) & 1u) << (7 - i);
    return r;
}

static int64_t powmod(int64_t b, uint32_t e)
{
    int64_t r = 1;
    b %= Q;
    while (e) {
        if (e & 1) r = (r * b) % Q;
        b = (b * b) % Q;
        e >>= 1;
    }
    return r;
}

void orc_init(void)
{
    if (zetas_ready) return;
    ZETAS[0] = 0;
    for (uint32_t k = 1; k < N; k++) {
        int64_t z = powmod(1753, brv8(k));
        if (z > (Q - 1) / 2) z -= Q;
        ZETAS[k] = (int32_t)z;
    }
    zetas_ready = 1;
}

const int32_t *orc_zetas(void)
{
    orc_init();
    return ZETAS;
}

/* canonical representative, the reference's compare rule util.cpp:98-101 */
int32_t orc_canon(int32_t x)
{
    int32_t r = x % Q;
    return r < 0 ? r + Q : r;
}

void orc_canon_poly(int32_t a[N])
{
    for (int i = 0; i < N; i++) a[i] = orc_canon(a[i]);
}

/* ------------------------------------------------------------------ */
/* H2: ntt -- ref_ntt.cpp:28-47.  In-place Cooley-Tukey, 8 layers,       */
/* len 128..1, twiddle index pre-incremented, C signed %.               */
/* ------------------------------------------------------------------ */
void orc_ntt(int32_t a[N])
{
    orc_init();
    unsigned k = 0;
    for (unsigned len = N / 2; len >= 1; len >>= 1) {
        for (unsigned start = 0; start < N; start += 2 * len) {
            int64_t zeta = ZETAS[++k];
            for (unsigned j = start; j < start + len; j++) {
                int32_t t = (int32_t)((zeta * a[j + len]) % Q);
                int32_t u = a[j];
                a[j + len] = (u - t) % Q;
                a[j] = (u + t) % Q;
            }
        }
    }
}

/* H3: invntt -- ref_ntt.cpp:59-87.  Gentleman-Sande, len 1..128, twiddle */
/* -zetas[--k], final scaling by f = 256^-1 mod q = 8347681 (:64).        */
void orc_invntt(int32_t a[N])
{
    orc_init();
    unsigned k = N;
    for (unsigned len = 1; len < N; len <<= 1) {
        for (unsigned start = 0; start < N; start += 2 * len) {
            int64_t zeta = -(int64_t)ZETAS[--k];
            for (unsigned j = start; j < start + len; j++) {
                int32_t t = a[j];
                a[j] = (t + a[j + len]) % Q;
                int32_t w = (t - a[j + len]) % Q;
                a[j + len] = (int32_t)((zeta * w) % Q);
            }
        }
    }
    const int64_t f = 8347681;
    for (unsigned j = 0; j < N; j++) a[j] = (int32_t)((f * a[j]) % Q);
}

/* H4: pointwise_barrett -- ref_ntt.cpp:49-57 (c may alias a).            */
void orc_pointwise(int32_t c[N], const int32_t a[N], const int32_t b[N])
{
    for (unsigned i = 0; i < N; i++) c[i] = (int32_t)(((int64_t)a[i] * b[i]) % Q);
}

/* ------------------------------------------------------------------ */
/* H5: radix-2x2 -- ref_ntt2x2.cpp:37-82 (forward), :100-145 (inverse).  */
/* Two layers per pass; forward twiddles z[k1], z[2k1], z[2k1+1] with    */
/* k1 = (N+i)>>l; inverse halves after every butterfly (:91-98).         */
/* ------------------------------------------------------------------ */
static inline void ct(int32_t *x, int32_t *y, int64_t z)
{
    int32_t t = (int32_t)((z * *y) % Q);
    *y = (*x - t) % Q;
    *x = (*x + t) % Q;
}

void orc_ntt2x2(int32_t a[N])
{
    orc_init();
    for (int l = LOGN; l > 0; l -= 2) {
        unsigned len = 1u << (l - 2);
        for (unsigned i = 0; i < N; i += 1u << l) {
            unsigned k1 = (N + i) >> l;
            int64_t za = ZETAS[k1], zb0 = ZETAS[2 * k1], zb1 = ZETAS[2 * k1 + 1];
            for (unsigned j = i; j < i + len; j++) {
                int32_t *p0 = &a[j], *p1 = &a[j + len], *p2 = &a[j + 2 * len], *p3 = &a[j + 3 * len];
                ct(p0, p2, za);
                ct(p1, p3, za);
                ct(p0, p1, zb0);
                ct(p2, p3, zb1);
            }
        }
    }
}

static inline int32_t half(int32_t t) /* ref_ntt2x2.cpp:91 */
{
    return (t & 1) ? ((t >> 1) + (Q + 1) / 2) : (t >> 1);
}

static inline void gs_half(int32_t *x, int32_t *y, int64_t z)
{
    int32_t d = half((*x - *y) % Q);
    *x = half((*x + *y) % Q);
    *y = (int32_t)((d * z) % Q);
}

void orc_invntt2x2(int32_t a[N])
{
    orc_init();
    for (int l = 0; l < LOGN; l += 2) {
        unsigned len = 1u << l;
        for (unsigned i = 0; i < N; i += 1u << (l + 2)) {
            unsigned ka = ((N - i / 2) >> l) - 1;
            unsigned kb = ((N - i / 2) >> (l + 1)) - 1;
            int64_t za0 = -(int64_t)ZETAS[ka], za1 = -(int64_t)ZETAS[ka - 1], zb = -(int64_t)ZETAS[kb];
            for (unsigned j = i; j < i + len; j++) {
                int32_t *p0 = &a[j], *p1 = &a[j + len], *p2 = &a[j + 2 * len], *p3 = &a[j + 3 * len];
                gs_half(p0, p1, za0);
                gs_half(p2, p3, za1);
                gs_half(p0, p2, zb);
                gs_half(p1, p3, zb);
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* H6: the hardware-model API on `bram` (config.h:29-51: 64 rows x 4      */
/* coefficients, row r = coefficients 4r..4r+3, util.cpp:61-72).         */
/* Restated SEMANTICALLY: `mapping` is an address translation applied to */
/* every row access (address_encoder_decoder.cpp:34-55); the in-place    */
/* row schedule of ntt2x2_fwdntt.cpp:32-157 leaves logical output row r  */
/* at logical address (r%4)*16 + r/4, that of ntt2x2_invntt.cpp:38-161   */
/* at (r%16)*4 + r/16.  Values come back canonical in [0,q) (the model   */
/* returns (-q,q); the reference compares canonically, util.cpp:98-112). */
/* ------------------------------------------------------------------ */
enum { ORC_NATURAL = 0, ORC_AFTER_NTT = 1, ORC_AFTER_INVNTT = 2 };

unsigned orc_resolve_address(int mapping, unsigned addr)
{
    switch (mapping) {
    case ORC_AFTER_INVNTT: return (addr % 16) * 4 + addr / 16;
    case ORC_AFTER_NTT:    return (addr % 4) * 16 + addr / 4;
    default:               return addr;
    }
}

static void bram_gather(int32_t poly[N], const int32_t *ram, int mapping)
{
    for (unsigned r = 0; r < 64; r++)
        memcpy(&poly[4 * r], &ram[4 * orc_resolve_address(mapping, r)], 4 * sizeof(int32_t));
}

void orc_bram_fwdntt(int32_t *ram /*[64][4]*/, int mapping)
{
    int32_t poly[N];
    bram_gather(poly, ram, mapping);
    orc_ntt2x2(poly);
    for (unsigned r = 0; r < 64; r++) {
        unsigned phys = orc_resolve_address(mapping, orc_resolve_address(ORC_AFTER_NTT, r));
        for (int c = 0; c < 4; c++) ram[4 * phys + c] = orc_canon(poly[4 * r + c]);
    }
}

void orc_bram_invntt(int32_t *ram, int mapping)
{
    int32_t poly[N];
    bram_gather(poly, ram, mapping);
    orc_invntt2x2(poly);
    for (unsigned r = 0; r < 64; r++) {
        unsigned phys = orc_resolve_address(mapping, orc_resolve_address(ORC_AFTER_INVNTT, r));
        for (int c = 0; c < 4; c++) ram[4 * phys + c] = orc_canon(poly[4 * r + c]);
    }
}

/* ntt2x2_mul.cpp:33-59: ram[map(l)][k] *= mul_ram[l][k] (the lane shuffle */
/* :48-51 cancels inside the butterfly's MUL mode, butterfly_unit.h:153-187) */
void orc_bram_mul(int32_t *ram, const int32_t *mul_ram, int mapping)
{
    for (unsigned l = 0; l < 64; l++) {
        unsigned phys = orc_resolve_address(mapping, l);
        for (int c = 0; c < 4; c++)
            ram[4 * phys + c] = orc_canon((int32_t)(((int64_t)ram[4 * phys + c] * mul_ram[4 * l + c]) % Q));
    }
}

/* ------------------------------------------------------------------ */
/* H7: RTL arithmetic spec.                                             */
/* Barrett_8380417.v:146-283: quo = ((x>>22)*8396807)>>24,               */
/* rem = (x[23:0] - (quo*q)[23:0]) mod 2^24, one conditional -q.  x<2^46. */
/* ------------------------------------------------------------------ */
uint32_t orc_barrett_rtl(uint64_t x)
{
    uint64_t quo = (((x >> 22) & 0xFFFFFFu) * 8396807ull) >> 24;
    uint32_t rem = ((uint32_t)(x & 0xFFFFFFu) - (uint32_t)((quo * Q) & 0xFFFFFFu)) & 0xFFFFFFu;
    uint32_t rmq = (rem - Q) & 0xFFFFFFu;
    return (rmq & 0x800000u) ? rem : rmq;
}

/* butterfly.v:27-250 op set on canonical residues.  mode: 0 FWD (CT),    */
/* 1 INV (GS with zeta -> q-zeta :186, then halve :214-222), 2 MULT =     */
/* multiply-accumulate acc + a*b (:144-150,224-230), 3 ADD, 4 SUB.        */
static uint32_t halve_u(uint32_t t) { return (t & 1) ? (t >> 1) + (Q + 1) / 2 : (t >> 1); }

void orc_butterfly_rtl(int mode, uint32_t aj, uint32_t ajlen, uint32_t zeta, uint32_t acc,
                       uint32_t *bj, uint32_t *bjlen)
{
    switch (mode) {
    case 0: {
        uint32_t t = orc_barrett_rtl((uint64_t)ajlen * zeta);
        uint32_t s = aj + t; if (s >= Q) s -= Q;
        uint32_t d = (t > aj) ? aj + Q - t : aj - t;
        *bj = s; *bjlen = d;
    } break;
    case 1: {
        uint32_t s = aj + ajlen; if (s >= Q) s -= Q;
        uint32_t d = (ajlen > aj) ? aj + Q - ajlen : aj - ajlen;
        uint32_t zi = (zeta == 0) ? 0 : Q - zeta;
        uint32_t m = orc_barrett_rtl((uint64_t)d * zi);
        *bj = halve_u(s); *bjlen = halve_u(m);
    } break;
    case 2: {
        uint32_t m = orc_barrett_rtl((uint64_t)aj * ajlen);
        uint32_t s = acc + m; if (s >= Q) s -= Q;
        *bj = 0; *bjlen = s;
    } break;
    case 3: {
        uint32_t s = aj + ajlen; if (s >= Q) s -= Q;
        *bj = 0; *bjlen = s;
    } break;
    default: {
        uint32_t d = (ajlen > aj) ? aj + Q - ajlen : aj - ajlen;
        *bj = 0; *bjlen = d;
    } break;
    }
}

/* The 2x2 unit as the reference's C++ hardware model composes it          */
/* (hardware_code/butterfly_unit.h:112-196 `buttefly_circuit`, the model of */
/* butterfly2x2.v): two butterflies (a,b | w[0]) (c,d | w[1]), the lane       */
/* exchange b <-> c, two butterflies (a',b' | w[2]) (c',d' | w[3]).  In       */
/* MUL mode the unit multiplies four lanes by four constants: the first two  */
/* multipliers take lanes b, d; lanes a, c are switched onto the second two  */
/* multipliers (:153-158) and switched back at the output (:180-187), so     */
/* out = {w[2] a, w[0] b, w[3] c, w[1] d}.  Built here from the RTL butterfly */
/* op set above on canonical residues; mode numbering = enum OPERATION       */
/* (config.h:39-44): 0 forward, 1 inverse, 2 mul.  The C++ unit's inverse     */
/* butterfly computes (ajlen - aj) * zeta (butterfly_unit.h:48-52), the RTL's */
/* (aj - ajlen) * (q - zeta) (butterfly.v:186): the same product, so the     */
/* same w feeds both.                                                        */
void orc_butterfly_circuit(int mode, const int32_t in[4], const int32_t w[4], int32_t out[4])
{
    uint32_t a = (uint32_t)orc_canon(in[0]), b = (uint32_t)orc_canon(in[1]), c = (uint32_t)orc_canon(in[2]), d = (uint32_t)orc_canon(in[3]);
    uint32_t z[4];
    for (int i = 0; i < 4; i++) z[i] = (uint32_t)orc_canon(w[i]);
    uint32_t a1, b1, c1, d1, a3, b3, c3, d3, unused;
    if (mode == 2) {
        orc_butterfly_rtl(2, b, z[0], 0, 0, &unused, &b1);       /* first pair of multipliers: lanes b, d */
        orc_butterfly_rtl(2, d, z[1], 0, 0, &unused, &d1);
        orc_butterfly_rtl(2, a, z[2], 0, 0, &unused, &b3);       /* lanes a, c switched onto the second pair */
        orc_butterfly_rtl(2, c, z[3], 0, 0, &unused, &d3);
        out[0] = (int32_t)b3; out[1] = (int32_t)b1; out[2] = (int32_t)d3; out[3] = (int32_t)d1;
        return;
    }
    orc_butterfly_rtl(mode, a, b, z[0], 0, &a1, &b1);
    orc_butterfly_rtl(mode, c, d, z[1], 0, &c1, &d1);
    orc_butterfly_rtl(mode, a1, c1, z[2], 0, &a3, &b3);          /* lane exchange: (a1, c1) and (b1, d1) */
    orc_butterfly_rtl(mode, b1, d1, z[3], 0, &c3, &d3);
    out[0] = (int32_t)a3; out[1] = (int32_t)b3; out[2] = (int32_t)c3; out[3] = (int32_t)d3;
}

/* twiddle_resolver.v:106-130 (forward) / :87-105 (inverse): the ROM       */
/* addresses of the (first-layer, first-layer, second-layer x2) twiddles  */
/* of 2x2 step m of stage s (s = 0..3).                                    */
void orc_twiddle_addrs(int inverse, unsigned s, unsigned m, unsigned out[4])
{
    if (!inverse) {
        unsigned k1 = (1u << (2 * s)) + m;
        out[0] = k1; out[1] = k1; out[2] = 2 * k1; out[3] = 2 * k1 + 1;
    } else {
        unsigned ka = (N >> (2 * s)) - 1 - 2 * m;
        unsigned kb = (N >> (2 * s + 1)) - 1 - m;
        out[0] = ka; out[1] = ka - 1; out[2] = kb; out[3] = kb;
    }
}

/* ------------------------------------------------------------------ */
/* Dilithium parameter table -- combined_top.v:518-552, norm_check.v:44-51, */
/* gen_c.v:107-119, usehint.v:59-62.                                      */
/* ------------------------------------------------------------------ */
typedef struct {
    int K, L, eta, tau, omega, beta;
    int32_t gamma1, gamma2;
} orc_params;

int orc_get_params(int level, orc_params *p)
{
    switch (level) {
    case 2: *p = (orc_params){4, 4, 2, 39, 80, 78, 1 << 17, (Q - 1) / 88}; return 0;
    case 3: *p = (orc_params){6, 5, 4, 49, 55, 196, 1 << 19, (Q - 1) / 32}; return 0;
    case 5: *p = (orc_params){8, 7, 2, 60, 75, 120, 1 << 19, (Q - 1) / 32}; return 0;
    default: return -1;
    }
}

/* Decompose, round-3 formulas, a canonical in [0,q):    */
/* a1 = HighBits, *a0 = LowBits centred in (-gamma2, gamma2].             */
int32_t orc_decompose(int level, int32_t a, int32_t *a0)
{
    int32_t a1 = (a + 127) >> 7;
    int32_t g2;
    if (level == 2) {
        g2 = (Q - 1) / 88;
        a1 = (a1 * 11275 + (1 << 23)) >> 24;
        a1 ^= ((43 - a1) >> 31) & a1;
    } else {
        g2 = (Q - 1) / 32;
        a1 = (a1 * 1025 + (1 << 21)) >> 22;
        a1 &= 15;
    }
    *a0 = a - a1 * 2 * g2;
    *a0 -= (((Q - 1) / 2 - *a0) >> 31) & Q;
    return a1;
}

/* Decompose the way the RTL does it: threshold map decomp_map1.v:37-171, */
/* low part coeff_decomposer.v:70-89 (returned as an unsigned residue).   */
int32_t orc_decompose_rtl(int level, int32_t a, int32_t *a0_unsigned)
{
    int32_t a1 = 0;
    if (level == 2) {
        /* thresholds 95233 + 190464*(i-1), i = 1..44; i = 44 wraps to 0 */
        for (int i = 44; i >= 1; i--)
            if (a >= 95233 + 190464 * (i - 1)) { a1 = (i == 44) ? 0 : i; break; }
    } else {
        for (int i = 16; i >= 1; i--)
            if (a >= 261889 + 523776 * (i - 1)) { a1 = (i == 16) ? 0 : i; break; }
    }
    int64_t a0 = (int64_t)a - (level == 2 ? (int64_t)a1 * 190464 : (int64_t)a1 * 523776);
    if (a0 > (Q - 1) / 2) a0 -= Q;
    if (a0 < 0) a0 += Q;
    *a0_unsigned = (int32_t)a0;
    return a1;
}

/* UseHint on (unsigned r0, r1) exactly as usehint.v:140-159 writes it.    */
int32_t orc_use_hint_rtl(int level, int32_t r0_unsigned, int32_t r1, int hint)
{
    if (!hint) return r1;
    if (level == 2) {
        if (r0_unsigned > (Q - 1) / 88 || r0_unsigned == 0) return (r1 == 0) ? 43 : r1 - 1;
        return (r1 == 43) ? 0 : r1 + 1;
    }
    if (r0_unsigned > (Q - 1) / 32 || r0_unsigned == 0) return (r1 == 0) ? 15 : r1 - 1;
    return (r1 == 15) ? 0 : r1 + 1;
}

/* UseHint, round-3 formula form.                                         */
int32_t orc_use_hint(int level, int32_t a, int hint)
{
    int32_t a0, a1 = orc_decompose(level, a, &a0);
    if (!hint) return a1;
    if (level == 2) return (a0 > 0) ? ((a1 == 43) ? 0 : a1 + 1) : ((a1 == 0) ? 43 : a1 - 1);
    return (a0 > 0) ? ((a1 + 1) & 15) : ((a1 - 1) & 15);
}

/* MakeHint on unsigned residue a0 -- makehint.v:98-99.                    */
int orc_make_hint(int level, int32_t a0_unsigned, int32_t a1)
{
    int32_t g2 = (level == 2) ? (Q - 1) / 88 : (Q - 1) / 32;
    int none = (a0_unsigned <= g2) || (a0_unsigned > Q - g2) || (a0_unsigned == Q - g2 && a1 == 0);
    return !none;
}

/* norm_check.v:84-105: reject when bound <= x <= q - bound (x unsigned). */
int orc_norm_reject(int32_t x_unsigned, int32_t bound)
{
    return x_unsigned >= bound && x_unsigned <= Q - bound;
}

/* ------------------------------------------------------------------ */
/* H9: mat-vec w = INTT(sum_l A[k][l] o NTT(y_l)) -- combined_top.v:1850-1933 */
/* (NTT_Y, MULT_A_Y with acc = 0 at l == 0 :1360/:1889, NTTI_W).           */
/* A row-major [K][L][256] in the NTT domain; all outputs canonical.      */
/* ------------------------------------------------------------------ */
static void mac_poly(int32_t acc[N], const int32_t a[N], const int32_t b[N])
{
    for (int i = 0; i < N; i++)
        acc[i] = (int32_t)(((int64_t)acc[i] + (int64_t)orc_canon(a[i]) * orc_canon(b[i])) % Q);
}

void orc_matvec(int K, int L, const int32_t *A, const int32_t *y, int32_t *w)
{
    int32_t yh[8][N];
    for (int l = 0; l < L; l++) {
        memcpy(yh[l], y + (size_t)l * N, sizeof(yh[l]));
        orc_ntt(yh[l]);
        orc_canon_poly(yh[l]);
    }
    for (int k = 0; k < K; k++) {
        int32_t acc[N] = {0};
        for (int l = 0; l < L; l++) mac_poly(acc, A + ((size_t)k * L + l) * N, yh[l]);
        orc_invntt(acc);
        orc_canon_poly(acc);
        memcpy(w + (size_t)k * N, acc, sizeof(acc));
    }
}

/* ------------------------------------------------------------------ */
/* H8: verify core -- combined_top.v:1207-1469:                          */
/* VY_NTT_Z, VY_NTT_T1 (t1 already scaled by 2^13, decoder.v:96-100),     */
/* VY_NTT_C, VY_MULT_AZ (MAC), VY_MULT_CT1, VY_SUB_AZ_CT1, VY_INTT,       */
/* VY_GENW1 = decompose + UseHint.  Inputs: z [L][256] canonical,         */
/* c [256] canonical (+-1 -> 1 / q-1), t1 [K][256] (10-bit, UNscaled),    */
/* h [K][256] 0/1 bytes; output w1 [K][256] bytes.                        */
/* ------------------------------------------------------------------ */
void orc_verify_core(int level, const int32_t *A, const int32_t *z, const int32_t *c,
                     const int32_t *t1, const uint8_t *h, uint8_t *w1, int32_t *w_out /*nullable*/)
{
    orc_params p;
    if (orc_get_params(level, &p)) return;
    int32_t zh[8][N], ch[N];
    for (int l = 0; l < p.L; l++) {
        memcpy(zh[l], z + (size_t)l * N, sizeof(zh[l]));
        orc_ntt(zh[l]);
        orc_canon_poly(zh[l]);
    }
    memcpy(ch, c, sizeof(ch));
    orc_ntt(ch);
    orc_canon_poly(ch);
    for (int k = 0; k < p.K; k++) {
        int32_t acc[N] = {0}, t[N];
        for (int l = 0; l < p.L; l++) mac_poly(acc, A + ((size_t)k * p.L + l) * N, zh[l]);
        for (int i = 0; i < N; i++) t[i] = (int32_t)(((int64_t)t1[(size_t)k * N + i] << 13) % Q);
        orc_ntt(t);
        orc_canon_poly(t);
        for (int i = 0; i < N; i++) {
            int64_t ct1 = ((int64_t)ch[i] * t[i]) % Q;
            acc[i] = (int32_t)((acc[i] - ct1 + Q) % Q);
        }
        orc_invntt(acc);
        orc_canon_poly(acc);
        if (w_out) memcpy(w_out + (size_t)k * N, acc, sizeof(acc));
        for (int i = 0; i < N; i++)
            w1[(size_t)k * N + i] = (uint8_t)orc_use_hint(level, acc[i], h[(size_t)k * N + i]);
    }
}

/* ------------------------------------------------------------------ */
/* H10: sign inner loop.                                                 */
/* phase 1 (operator 0, FSM1: combined_top.v:1830-1933 + DECOMP :1946):   */
/*   w = INTT(A o NTT(y)); (w1, w0) = Decompose(w).  w0 is returned as an */
/*   unsigned residue in [0,q), the way the RTL keeps it.                 */
/* phase 2 (operator 1, FSM2: :1981-2229):                                */
/*   z = y + INTT(c^ o s1^) with reject ||z|| >= gamma1-beta (:2088-2101); */
/*   r0 = w0 - INTT(c^ o s2^) with reject >= gamma2-beta (:2163-2177);    */
/*   ct0 = INTT(c^ o t0^) with reject >= gamma2 (:2133-2145);             */
/*   h = MakeHint(r0 + ct0, w1), reject when #h > omega (:2189, makehint.v). */
/* flags bit0 z-norm, bit1 r0-norm, bit2 ct0-norm, bit3 hint count.       */
/* ------------------------------------------------------------------ */
void orc_sign_phase1(int level, const int32_t *A, const int32_t *y, uint8_t *w1, int32_t *w0)
{
    orc_params p;
    if (orc_get_params(level, &p)) return;
    int32_t w[8 * N];
    orc_matvec(p.K, p.L, A, y, w);
    for (int i = 0; i < p.K * N; i++) {
        int32_t a0, a1 = orc_decompose(level, w[i], &a0);
        w1[i] = (uint8_t)a1;
        w0[i] = a0 < 0 ? a0 + Q : a0;
    }
}

int orc_sign_phase2(int level, const int32_t *c, const int32_t *y, const int32_t *w0,
                    const uint8_t *w1, const int32_t *s1hat, const int32_t *s2hat,
                    const int32_t *t0hat, int32_t *z, uint8_t *h)
{
    orc_params p;
    if (orc_get_params(level, &p)) return -1;
    int flags = 0;
    int32_t ch[N], t[N];
    memcpy(ch, c, sizeof(ch));
    orc_ntt(ch);
    orc_canon_poly(ch);
    for (int l = 0; l < p.L; l++) {
        for (int i = 0; i < N; i++)
            t[i] = (int32_t)(((int64_t)ch[i] * orc_canon(s1hat[(size_t)l * N + i])) % Q);
        orc_invntt(t);
        for (int i = 0; i < N; i++) {
            int32_t v = orc_canon((int32_t)(((int64_t)orc_canon(t[i]) + orc_canon(y[(size_t)l * N + i])) % Q));
            z[(size_t)l * N + i] = v;
            if (orc_norm_reject(v, p.gamma1 - p.beta)) flags |= 1;
        }
    }
    int nh = 0;
    for (int k = 0; k < p.K; k++) {
        int32_t cs2[N], ct0[N];
        for (int i = 0; i < N; i++) {
            cs2[i] = (int32_t)(((int64_t)ch[i] * orc_canon(s2hat[(size_t)k * N + i])) % Q);
            ct0[i] = (int32_t)(((int64_t)ch[i] * orc_canon(t0hat[(size_t)k * N + i])) % Q);
        }
        orc_invntt(cs2);
        orc_invntt(ct0);
        for (int i = 0; i < N; i++) {
            int32_t r0 = orc_canon(orc_canon(w0[(size_t)k * N + i]) - orc_canon(cs2[i]));
            int32_t u = orc_canon(ct0[i]);
            if (orc_norm_reject(r0, p.gamma2 - p.beta)) flags |= 2;
            if (orc_norm_reject(u, p.gamma2)) flags |= 4;
            int32_t s = orc_canon((int32_t)(((int64_t)r0 + u) % Q));
            int hb = orc_make_hint(level, s, w1[(size_t)k * N + i]);
            h[(size_t)k * N + i] = (uint8_t)hb;
            nh += hb;
        }
    }
    if (nh > p.omega) flags |= 8;
    return flags;
}

/* ------------------------------------------------------------------ */
/* batch drivers + timers for bench.py's cpu_baseline leg                */
/* ------------------------------------------------------------------ */
void orc_ntt_batch(int32_t *a, size_t n)    { for (size_t i = 0; i < n; i++) orc_ntt(a + i * N); }
void orc_invntt_batch(int32_t *a, size_t n) { for (size_t i = 0; i < n; i++) orc_invntt(a + i * N); }
void orc_canon_batch(int32_t *a, size_t n)  { for (size_t i = 0; i < n * N; i++) a[i] = orc_canon(a[i]); }
void orc_ntt2x2_batch(int32_t *a, size_t n)    { for (size_t i = 0; i < n; i++) orc_ntt2x2(a + i * N); }
void orc_invntt2x2_batch(int32_t *a, size_t n) { for (size_t i = 0; i < n; i++) orc_invntt2x2(a + i * N); }
void orc_pointwise_batch(int32_t *c, const int32_t *a, const int32_t *b, size_t n)
{
    for (size_t i = 0; i < n; i++) orc_pointwise(c + i * N, a + i * N, b + i * N);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* time `reps` sweeps of fn over n polynomials; fn may be one of ours or a */
/* function pointer into oracle/_ref (the compiled reference).            */
double orc_time_poly_fn(void (*fn)(int32_t *), int32_t *a, size_t n, int reps)
{
    double t0 = now_s();
    for (int r = 0; r < reps; r++)
        for (size_t i = 0; i < n; i++) fn(a + i * N);
    return now_s() - t0;
}

double orc_time_verify_core(int level, const int32_t *A, const int32_t *z, const int32_t *c,
                            const int32_t *t1, const uint8_t *h, uint8_t *w1, size_t n, int shared_pk)
{
    orc_params p;
    if (orc_get_params(level, &p)) return -1.0;
    size_t szA = (size_t)p.K * p.L * N, szz = (size_t)p.L * N, szk = (size_t)p.K * N;
    double t0 = now_s();
    for (size_t i = 0; i < n; i++)
        orc_verify_core(level, A + (shared_pk ? 0 : i * szA), z + i * szz, c + i * N,
                        t1 + (shared_pk ? 0 : i * szk), h + i * szk, w1 + i * szk, NULL);
    return now_s() - t0;
}

void orc_matvec_batch(int K, int L, const int32_t *A, const int32_t *y, int32_t *w, size_t n, int shared_A)
{
    for (size_t i = 0; i < n; i++)
        orc_matvec(K, L, A + (shared_A ? 0 : i * (size_t)K * L * N), y + i * (size_t)L * N, w + i * (size_t)K * N);
}

void orc_verify_core_batch(int level, const int32_t *A, const int32_t *z, const int32_t *c,
                           const int32_t *t1, const uint8_t *h, uint8_t *w1, size_t n, int shared_pk)
{
    (void)orc_time_verify_core(level, A, z, c, t1, h, w1, n, shared_pk);
}

/* batch forms of the sign phases (same per-item functions; the Python side may split a batch over threads) */
void orc_sign_phase1_batch(int level, const int32_t *A, const int32_t *y, uint8_t *w1, int32_t *w0, size_t n, int shared_key)
{
    orc_params p;
    if (orc_get_params(level, &p)) return;
    size_t szA = (size_t)p.K * p.L * N, szy = (size_t)p.L * N, szk = (size_t)p.K * N;
    for (size_t i = 0; i < n; i++)
        orc_sign_phase1(level, A + (shared_key ? 0 : i * szA), y + i * szy, w1 + i * szk, w0 + i * szk);
}

void orc_sign_phase2_batch(int level, const int32_t *c, const int32_t *y, const int32_t *w0, const uint8_t *w1,
                           const int32_t *s1hat, const int32_t *s2hat, const int32_t *t0hat, int32_t *z, uint8_t *h,
                           int32_t *flags, size_t n, int shared_key)
{
    orc_params p;
    if (orc_get_params(level, &p)) return;
    size_t szl = (size_t)p.L * N, szk = (size_t)p.K * N;
    for (size_t i = 0; i < n; i++) {
        size_t j = shared_key ? 0 : i;
        flags[i] = orc_sign_phase2(level, c + i * N, y + i * szl, w0 + i * szk, w1 + i * szk, s1hat + j * szl,
                                   s2hat + j * szk, t0hat + j * szk, z + i * szl, h + i * szk);
    }
}

/* ------------------------------------------------------------------ */
/* exhaustive self-checks used by tests/test_oracle_*.py                 */
/* ------------------------------------------------------------------ */
/* formula Decompose/UseHint == RTL threshold-map Decompose/UseHint over all of [0,q) */
long orc_check_decompose_full(int level)
{
    long bad = 0;
    for (int32_t a = 0; a < Q; a++) {
        int32_t a0, a0u, a1 = orc_decompose(level, a, &a0);
        int32_t b1 = orc_decompose_rtl(level, a, &a0u);
        if (a1 != b1 || (a0 < 0 ? a0 + Q : a0) != a0u) bad++;
        for (int h = 0; h < 2; h++)
            if (orc_use_hint(level, a, h) != orc_use_hint_rtl(level, a0u, b1, h)) bad++;
    }
    return bad;
}

/* RTL Barrett == x mod q on n pseudo-random products of residues plus edge cases */
long orc_check_barrett(uint64_t seed, long n)
{
    long bad = 0;
    uint64_t s = seed;
    const uint64_t edge[] = {0, 1, Q - 1, Q, Q + 1, (uint64_t)(Q - 1) * (Q - 1), (1ull << 46) - 1,
                             (uint64_t)(Q - 1) * (Q - 2), 1ull << 22, (1ull << 22) - 1, 1ull << 45};
    for (unsigned i = 0; i < sizeof(edge) / sizeof(edge[0]); i++)
        if (orc_barrett_rtl(edge[i]) != edge[i] % Q) bad++;
    for (long i = 0; i < n; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        uint64_t a = (s >> 20) % Q;
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        uint64_t b = (s >> 20) % Q;
        if (orc_barrett_rtl(a * b) != (a * b) % Q) bad++;
    }
    return bad;
}
