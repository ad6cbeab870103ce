// kernels.hip -- hand-written CDNA4 (gfx950) kernels for the Dilithium NTT hot path.
//
// One 64-lane wavefront owns one polynomial (256 x int32 = 1 KiB): 4 coefficients per lane,
// butterflies in VGPRs, exchanges by cross-lane ops (ntt_core.hpp), signed Montgomery
// arithmetic on 32-bit multiplies (modarith.hpp).  No MFMA: this is 32-bit integer work bounded by HBM bandwidth
// and VALU issue.  Every polynomial crosses HBM exactly once per kernel.
//
// Reference behaviour implemented (file:line relative to GMUCERG/Dilithium):
//   ntt / invntt / pointwise_barrett          dilithium-256/reference_code/ref_ntt.cpp:28-87
//   ntt2x2_ref / invntt2x2_ref (same maps)    reference_code/ref_ntt2x2.cpp:37-145
//   ntt2x2_fwdntt / _invntt / _mul on `bram`  hardware_code/ntt2x2_{fwdntt,invntt,mul}.cpp,
//                                             address_encoder_decoder.cpp:34-55
//   butterfly op set FWD/INV/MAC/ADD/SUB      rtl_src/butterfly.v:27-250, butterfly2x2.v
//   verify core, mat-vec, sign inner loop     rtl_src/combined_top.v:1207-1469, :1850-1933,
//                                             :1946-2229; decompose/usehint/makehint/norm
//                                             coeff_decomposer.v, usehint.v:140-159,
//                                             makehint.v:98-99, norm_check.v:84-105
// Outputs are canonical residues in [0, q) (the RTL's convention, butterfly.v:194-195); the
// reference C++ returns (-q, q) and compares canonically (util.cpp:98-112).
#include "device_common.hpp"
#include "kernels.hpp"
#include "launch_util.hpp"
#include "ntt_core.hpp"

namespace dil {

// cache-policy A/B hooks of the standalone transforms (scripts/ab_verify.py --kind ntt): the strided 256-byte-per-instruction
// accesses (forward loads, inverse stores) can be built with the default policy instead of non-temporal
__device__ __forceinline__ int32_t ld_s(const int32_t* p) { return ld_nt(p); }
__device__ __forceinline__ void st_s(int32_t* p, int32_t v) { st_nt(p, v); }

// ---------------------------------------------------------------------------------------
// address translation of the hardware model's `bram` (address_encoder_decoder.cpp:34-55)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int resolve_row(int mapping, int addr)
{
    if (mapping == MAP_AFTER_INVNTT) return (addr & 15) * 4 + (addr >> 4);
    if (mapping == MAP_AFTER_NTT) return (addr & 3) * 16 + (addr >> 2);
    return addr;
}

// LAYOUT = LAYOUT_POLY : plain data_t[256] in reference order (ref_ntt.h API)
// LAYOUT = LAYOUT_BRAM : `bram` rows behind `mapping`; the transform leaves its output rows at
//                        the model's post-transform permutation (ntt2x2_test.cpp:55,76,129-132)
template <int LAYOUT>
__device__ __forceinline__ int fwd_in_off(int i, int mapping)
{
    if (LAYOUT == LAYOUT_POLY) return i;
    return 4 * resolve_row(mapping, i >> 2) + (i & 3);
}
template <int LAYOUT>
__device__ __forceinline__ int fwd_out_row_off(int row, int mapping)
{
    if (LAYOUT == LAYOUT_POLY) return 4 * row;
    return 4 * resolve_row(mapping, resolve_row(MAP_AFTER_NTT, row));
}
template <int LAYOUT>
__device__ __forceinline__ int inv_in_row_off(int row, int mapping)
{
    if (LAYOUT == LAYOUT_POLY) return 4 * row;
    return 4 * resolve_row(mapping, row);
}
template <int LAYOUT>
__device__ __forceinline__ int inv_out_off(int i, int mapping)
{
    if (LAYOUT == LAYOUT_POLY) return i;
    return 4 * resolve_row(mapping, resolve_row(MAP_AFTER_INVNTT, i >> 2)) + (i & 3);
}

// ---------------------------------------------------------------------------------------
// H2/H5/H6 forward NTT, batched, in place.  Persistent waves, grid-stride over polynomials,
// the next polynomial's loads are issued before the current one is transformed.
// HBM traffic: 1 KiB in (4 coalesced 256-B dword loads per wave) + 1 KiB out (one 1-KiB
// dwordx4 store per wave) per polynomial.
// ---------------------------------------------------------------------------------------
// Frozen in round 5.  Launch shape (1 / 2 / 4 waves per workgroup x 4 ... 16 workgroups per CU), s_setprio around the memory instructions,
// prefetch depth, chunking, the exchange policy and the product form were swept in rounds 1-4 (profiles/r01_tune_ntt.txt,
// r03e_tune_ntt2.txt, r04i_*, r04o_ab_ntt_shapes_prio.txt: all within +-1.5 % of this shape); round 5 measured the two memory-side
// forms left -- one dwordx4 load per lane with the transposition through LDS, and the same slot filled by global_load_lds_dwordx4 --
// and both lose with the arithmetic on (26.3 / 26.5 vs 25.95 us per 65536 polynomials, profiles/r05_ntt_x4.txt).  The kernel runs at
// 0.91 of its own loads and stores; the A/B hooks are gone.
constexpr int NTT_WPB = 4;          // waves per workgroup of the standalone transforms
template <int LAYOUT>
__global__ __launch_bounds__(64 * NTT_WPB) void ntt_fwd_kernel(int32_t* __restrict__ polys, size_t batch,
                                                       const uint32_t* __restrict__ tw_tab, int mapping)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * NTT_WPB + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * NTT_WPB;
    if (wave >= batch) return;
    int off[4];
#pragma unroll
    for (int m = 0; m < 4; m++) off[m] = fwd_in_off<LAYOUT>(lane + 64 * m, mapping);
    const int out_off = fwd_out_row_off<LAYOUT>(lane, mapping);
    TwRegs tw;
    const LaneMasks lm(lane);
    const LaneMasks& xp = lm;
    if (LAYOUT == LAYOUT_BRAM && mapping == MAP_AFTER_INVNTT) {
        // this mapping puts the lane's 4 inputs at 16 (lane>>2) + 4 m + (lane&3): 16-byte pieces 64 B apart for every
        // load instruction.  Read the wave's 1 KiB with one dwordx4 per lane instead and transpose inside each quad.
        int4 nx = ld_nt4(polys + wave * 256 + 4 * lane);
        tw.load(tw_tab, lane);
        for (size_t p = wave; p < batch; p += nwaves) {
            int32_t r[4] = {nx.x, nx.y, nx.z, nx.w};
            const size_t pn = p + nwaves;
            if (pn < batch) nx = ld_nt4(polys + pn * 256 + 4 * lane);
            xchg_10(r, lm);
            ntt_fwd_core(r, tw, xp);
            st_nt4(polys + p * 256 + out_off, canon_any(r[0]), canon_any(r[1]), canon_any(r[2]), canon_any(r[3]));
        }
        return;
    }
    int32_t nxt[4];
#pragma unroll
    for (int m = 0; m < 4; m++) nxt[m] = ld_s(polys + wave * 256 + off[m]);
    tw.load(tw_tab, lane);
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t r[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
        const size_t pn = p + nwaves;
        if (pn < batch) {
#pragma unroll
            for (int m = 0; m < 4; m++) nxt[m] = ld_s(polys + pn * 256 + off[m]);
        }
        ntt_fwd_core(r, tw, xp);
        const uint32_t o0 = canon_any(r[0]), o1 = canon_any(r[1]), o2 = canon_any(r[2]), o3 = canon_any(r[3]);
        st_nt4(polys + p * 256 + out_off, o0, o1, o2, o3);
    }
}

// H3/H5/H6 inverse NTT (x 256^-1), batched, in place.  Inputs in (-q, q) (or canonical).
template <int LAYOUT>
__global__ __launch_bounds__(64 * NTT_WPB) void ntt_inv_kernel(int32_t* __restrict__ polys, size_t batch,
                                                       const uint32_t* __restrict__ tw_tab, int mapping)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * NTT_WPB + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * NTT_WPB;
    if (wave >= batch) return;
    const int in_off = inv_in_row_off<LAYOUT>(lane, mapping);
    int off[4];
#pragma unroll
    for (int m = 0; m < 4; m++) off[m] = inv_out_off<LAYOUT>(lane + 64 * m, mapping);
    int4 nxt = ld_nt4(polys + wave * 256 + in_off);
    TwRegs tw;
    tw.load(tw_tab, lane);
    const LaneMasks lm(lane);
    const LaneMasks& xp = lm;
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t r[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
        const size_t pn = p + nwaves;
        if (pn < batch) nxt = ld_nt4(polys + pn * 256 + in_off);
        ntt_inv_core(r, tw, xp);
        if (LAYOUT == LAYOUT_BRAM && mapping == MAP_NATURAL) {
            // outputs of this (op, mapping) land at 16 (lane>>2) + 4 m + (lane&3): transpose inside each quad and
            // write the wave's 1 KiB as one dwordx4 per lane instead of four 16-byte-granular scatters
            xchg_10(r, lm);
            st_nt4(polys + p * 256 + 4 * lane, (int32_t)canon_small(r[0]), (int32_t)canon_small(r[1]), (int32_t)canon_small(r[2]),
                   (int32_t)canon_small(r[3]));
            continue;
        }
#pragma unroll
        for (int m = 0; m < 4; m++) st_s(polys + p * 256 + off[m], (int32_t)canon_small(r[m]));
    }
}

// ---------------------------------------------------------------------------------------
// H4 / butterfly.v MULT / ADD / SUB modes: element-wise ops on whole polynomials.
// 4 coefficients (16 B) per thread, grid-stride.  c may alias a (ntt2x2_test.cpp:102).
//   OP_MUL: c = a*b        OP_MAC: c = acc + a*b        OP_ADD: c = a+b       OP_SUB: c = a-b
// ---------------------------------------------------------------------------------------
// true product of two residues given as any int32 with |a|,|b| < 2^31 / ... (here: < 2^24):
// Montgomery-reduce the 64-bit product, then multiply by 2^64 mod q to cancel the 2^-32.
__device__ __forceinline__ int32_t mulmod_true(int32_t a, int32_t b) { return mont_tw(mont_mul(a, b), R2_WT, R2_WQ); }

template <int OP>
__global__ __launch_bounds__(256) void pointwise_kernel(int32_t* c, const int32_t* a, const int32_t* b,
                                                         const int32_t* acc, size_t nvec4)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec4; i += stride) {
        const int4 va = reinterpret_cast<const int4*>(a)[i];
        const int4 vb = reinterpret_cast<const int4*>(b)[i];
        const int32_t x[4] = {va.x, va.y, va.z, va.w};
        const int32_t y[4] = {vb.x, vb.y, vb.z, vb.w};
        uint32_t o[4];
        if (OP == OP_MUL) {
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = canon_small(mulmod_true(x[k], y[k]));
        } else if (OP == OP_MAC) {
            const int4 vc = reinterpret_cast<const int4*>(acc)[i];
            const int32_t z[4] = {vc.x, vc.y, vc.z, vc.w};
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = canon_any(z[k] + mulmod_true(x[k], y[k]));
        } else if (OP == OP_ADD) {
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = canon_any(x[k] + y[k]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) o[k] = canon_any(x[k] - y[k]);
        }
        reinterpret_cast<uint4*>(c)[i] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------------------
// The reference's polynomial product chain in ONE kernel: c = invntt(ntt(a) o ntt(b)) = a * b in Z_q[x] / (x^256 + 1)
// (ntt2x2_test.cpp:109-137 `polymul`: two forward transforms, the pointwise product, the inverse transform; ref_ntt.cpp:28-87).
// One wavefront owns a pair: both operands are transformed side by side in registers (ntt_fwd_core2: one set of twiddle reads, two
// dependency chains), the forward transform leaves a lane holding the four outputs the inverse transform wants in that lane, so
// the pointwise product is lane-local; one Montgomery product per coefficient (a^ b^ 2^-32, |.| < q: what the Gentleman-Sande
// sums need) and the "pipeline" flavour of the inverse table, whose final constant 2^32 / 256 cancels it.  HBM traffic per product:
// 2 KiB in, 1 KiB out -- against 9 KiB for the four launches of the chain (2 + 2 + 3 + 2).  c may alias a or b.
// Forward twiddles in registers (used twice per pair), inverse ones in LDS (8 KiB per workgroup, staged once).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * NTT_WPB) void polymul_kernel(int32_t* c, const int32_t* a, const int32_t* b, size_t batch,
                                                               const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_pipe_tab)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_inv[TW_TABLE_DWORDS];
    for (int i = threadIdx.x; i < TW_TABLE_DWORDS / 4; i += blockDim.x)
        reinterpret_cast<uint4*>(s_inv)[i] = reinterpret_cast<const uint4*>(inv_pipe_tab)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * NTT_WPB + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * NTT_WPB;
    if (wave >= batch) return;
    TwRegs twf;
    const TwLds twi{s_inv, lane};
    const LaneMasks lm(lane);
    int32_t na[4], nb[4];
#pragma unroll
    for (int m = 0; m < 4; m++) {
        na[m] = ld_s(a + wave * 256 + lane + 64 * m);
        nb[m] = ld_s(b + wave * 256 + lane + 64 * m);
    }
    twf.load(fwd_tab, lane);
    for (size_t p = wave; p < batch; p += nwaves) {
        int32_t ra[4] = {na[0], na[1], na[2], na[3]}, rb[4] = {nb[0], nb[1], nb[2], nb[3]};
        const size_t pn = p + nwaves;
        if (pn < batch) {
#pragma unroll
            for (int m = 0; m < 4; m++) {
                na[m] = ld_s(a + pn * 256 + lane + 64 * m);
                nb[m] = ld_s(b + pn * 256 + lane + 64 * m);
            }
        }
        ntt_fwd_core2(ra, rb, twf, lm);
        int32_t r[4];
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = mont_mul(ra[j], rb[j]);        // |ra|, |rb| < 9 q: |product| < 2^31 q
        ntt_inv_core(r, twi, lm);
#pragma unroll
        for (int m = 0; m < 4; m++) st_s(c + p * 256 + lane + 64 * m, (int32_t)canon_small(r[m]));
    }
}

// The transforms' memory traffic WITHOUT the arithmetic (bench.py `roofline.achievable`): the same persistent grid, the same
// prefetch distance, the same instructions to memory -- forward: four strided 256-byte dword loads, one 1-KiB dwordx4 store per
// wave and polynomial; inverse: mirrored -- so that the bench line carries, beside the 8 TB/s spec, what this access pattern
// reaches on the box it runs on.  Scrambles the buffer (a lane's four strided values leave as one row); scratch data only.
template <bool INVERSE>
__global__ __launch_bounds__(64 * NTT_WPB) void ntt_traffic_kernel(int32_t* __restrict__ polys, size_t batch)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * NTT_WPB + (threadIdx.x >> 6);
    const size_t nwaves = (size_t)gridDim.x * NTT_WPB;
    if (wave >= batch) return;
    if (!INVERSE) {
        int32_t nxt[4];
#pragma unroll
        for (int m = 0; m < 4; m++) nxt[m] = ld_s(polys + wave * 256 + lane + 64 * m);
        for (size_t p = wave; p < batch; p += nwaves) {
            const int32_t r[4] = {nxt[0], nxt[1], nxt[2], nxt[3]};
            const size_t pn = p + nwaves;
            if (pn < batch) {
#pragma unroll
                for (int m = 0; m < 4; m++) nxt[m] = ld_s(polys + pn * 256 + lane + 64 * m);
            }
            st_nt4(polys + p * 256 + 4 * lane, (uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
        }
    } else {
        int4 nx = ld_nt4(polys + wave * 256 + 4 * lane);
        for (size_t p = wave; p < batch; p += nwaves) {
            const int4 r = nx;
            const size_t pn = p + nwaves;
            if (pn < batch) nx = ld_nt4(polys + pn * 256 + 4 * lane);
            st_s(polys + p * 256 + lane, r.x);
            st_s(polys + p * 256 + lane + 64, r.y);
            st_s(polys + p * 256 + lane + 128, r.z);
            st_s(polys + p * 256 + lane + 192, r.w);
        }
    }
}

// ntt2x2_mul on `bram` (ntt2x2_mul.cpp:33-59): ram[map(l)][k] *= mul_ram[l][k]; one thread per row
__global__ __launch_bounds__(256) void bram_mul_kernel(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < batch * 64; i += stride) {
        const size_t p = i >> 6;
        const int l = (int)(i & 63);
        int4* dst = reinterpret_cast<int4*>(ram + p * 256 + 4 * resolve_row(mapping, l));
        const int4 va = *dst;
        const int4 vb = *reinterpret_cast<const int4*>(mul_ram + p * 256 + 4 * l);
        int4 o;
        o.x = (int32_t)canon_small(mulmod_true(va.x, vb.x));
        o.y = (int32_t)canon_small(mulmod_true(va.y, vb.y));
        o.z = (int32_t)canon_small(mulmod_true(va.z, vb.z));
        o.w = (int32_t)canon_small(mulmod_true(va.w, vb.w));
        *dst = o;
    }
}

// ---------------------------------------------------------------------------------------
// Host mailbox (kernels.hpp): one resident wave serving batch-of-one requests from pinned host memory.
// Same arithmetic as the batched kernels above -- ntt_fwd_core / ntt_inv_core on 4 coefficients per lane, the `bram` address
// maps, mulmod_true -- on ONE polynomial that never touches device memory: loads and stores go to the mailbox over PCIe.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mb_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void mb_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// the request header (sequence number, op, mapping in ONE 32-bit word: kernels.hpp mb_header) in one system-scope load = one PCIe read
// that cannot be torn; wave-uniform
__device__ __forceinline__ uint32_t mb_load_header(const Mailbox* mb)
{
    uint32_t h;
    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(h) : "v"(&mb->req_seq) : "memory");
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
}

__global__ __launch_bounds__(64) void mailbox_kernel(Mailbox* mb, const uint32_t* __restrict__ fwd_tab, const uint32_t* __restrict__ inv_tab,
                                                     uint32_t last_done, uint64_t idle_ticks, uint64_t max_resident_ticks)
{
    const int lane = threadIdx.x & 63;
    TwRegs twf, twi;
    twf.load(fwd_tab, lane);
    twi.load(inv_tab, lane);
    const LaneMasks lm(lane);
    uint32_t done = last_done, served = 0;
    const uint64_t t_start = wall_clock64();            // 100 MHz, independent of the shader clock
    uint64_t t_last = t_start;
    for (;;) {
        uint32_t hdr = mb_load_header(mb);
        if (hdr == done) {
            // Retire after `idle_ticks` without a request -- and, busy or not, after `max_resident_ticks` in all: a caller issuing
            // requests back to back would otherwise keep the wave resident for ever and starve every other thread's device-wide
            // synchronisation (hipFree, hipMalloc, hipDeviceSynchronize); the next call relaunches it (~40 us, once per residency).
            const uint64_t now = wall_clock64();
            if (now - t_last < idle_ticks && now - t_start < max_resident_ticks) continue;
            // retire -- unless a request slips in: announce, look once more (a PCIe read cannot overtake the posted write before
            // it, so either this read sees the request or the host, which reads `state` after posting, sees EXITING / DEAD)
            if (lane == 0) mb_store(&mb->state, MB_EXITING);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            hdr = mb_load_header(mb);
            if (hdr == done) {
                if (lane == 0) {
                    mb_store(&mb->served, served);
                    mb_store(&mb->state, MB_DEAD);
                }
                return;
            }
            if (lane == 0) mb_store(&mb->state, MB_ALIVE);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");     // the payload was written before the header
        const uint32_t op = (hdr >> 2) & 63u;
        const int mapping = (int)(hdr & 3u);
        if (op == MB_QUIT) {
            if (lane == 0) {
                mb_store(&mb->done_seq, hdr);
                mb_store(&mb->state, MB_DEAD);
            }
            return;
        }
        int32_t r[4];
        if (op == MB_FWD || op == MB_BRAM_FWD) {
            const bool br = op == MB_BRAM_FWD;
#pragma unroll
            for (int m = 0; m < 4; m++) r[m] = mb->in0[br ? fwd_in_off<LAYOUT_BRAM>(lane + 64 * m, mapping) : lane + 64 * m];
            ntt_fwd_core(r, twf, lm);
            int32_t* o = mb->out + (br ? fwd_out_row_off<LAYOUT_BRAM>(lane, mapping) : 4 * lane);
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (int32_t)canon_any(r[j]);
        } else if (op == MB_INV || op == MB_BRAM_INV) {
            const bool br = op == MB_BRAM_INV;
            const int32_t* in = mb->in0 + (br ? inv_in_row_off<LAYOUT_BRAM>(lane, mapping) : 4 * lane);
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = in[j];
            ntt_inv_core(r, twi, lm);
#pragma unroll
            for (int m = 0; m < 4; m++) mb->out[br ? inv_out_off<LAYOUT_BRAM>(lane + 64 * m, mapping) : lane + 64 * m] = (int32_t)canon_small(r[m]);
        } else if (op == MB_POLYMUL) {                    // out = invntt(ntt(in0) o ntt(in1)): the reference's whole polymul chain in one request
            int32_t rb[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                r[m] = mb->in0[lane + 64 * m];
                rb[m] = mb->in1[lane + 64 * m];
            }
            ntt_fwd_core2(r, rb, twf, lm);
#pragma unroll
            for (int j = 0; j < 4; j++) r[j] = mulmod_true(r[j], rb[j]);
            ntt_inv_core(r, twi, lm);
#pragma unroll
            for (int m = 0; m < 4; m++) mb->out[lane + 64 * m] = (int32_t)canon_small(r[m]);
        } else {                                          // MB_PW_MUL: out = in0 o in1;  MB_BRAM_MUL: ram[map(l)] *= mul_ram[l] (ntt2x2_mul.cpp:33-59)
            const int row = op == MB_BRAM_MUL ? resolve_row(mapping, lane) : lane;
#pragma unroll
            for (int j = 0; j < 4; j++) mb->out[4 * row + j] = (int32_t)canon_small(mulmod_true(mb->in0[4 * row + j], mb->in1[4 * lane + j]));
        }
        served++;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) mb_store(&mb->done_seq, hdr);
        done = hdr;
        t_last = wall_clock64();
    }
}

// Shader-clock probe (bench.py): one lane reads the shader cycle counter (s_memtime) and the constant 100 MHz counter
// (s_memrealtime), dozes for `spin_ticks` of the latter and reads both again -- effective shader clock = d(cycles) / d(ticks) x 100 MHz
// over exactly the interval in which the kernels being timed run beside it on another stream.  (The chip clocks to its power budget:
// a VALU-dense kernel sustains 1.9-2.3 GHz, not the 2.4 GHz of the data sheet, MI355X_MICROARCH.md "DVFS give-back".)
__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t* out, uint64_t spin_ticks)
{
    if (threadIdx.x != 0) return;
    const uint64_t r0 = wall_clock64(), c0 = clock64();
    while (wall_clock64() - r0 < spin_ticks) __builtin_amdgcn_s_sleep(64);
    const uint64_t c1 = clock64(), r1 = wall_clock64();
    out[0] = c0; out[1] = c1; out[2] = r0; out[3] = r1;
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
hipError_t launch_clock_probe(uint64_t* out4, uint64_t spin_ticks, hipStream_t s)
{
    hipLaunchKernelGGL(clock_probe_kernel, 1, 64, 0, s, out4, spin_ticks);
    return hipGetLastError();
}

hipError_t launch_mailbox(Mailbox* mb_dev, uint32_t last_done, uint64_t idle_ticks, uint64_t max_resident_ticks, const Tables& t, hipStream_t s)
{
    hipLaunchKernelGGL(mailbox_kernel, 1, 64, 0, s, mb_dev, t.fwd, t.inv, last_done, idle_ticks, max_resident_ticks);
    return hipGetLastError();
}

hipError_t launch_ntt(bool inverse, int layout, int mapping, int32_t* polys, size_t batch,
                      const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    const int grid = grid_for((batch + NTT_WPB - 1) / NTT_WPB, t.num_cus * t.ntt_blocks_per_cu * 4 / NTT_WPB);
    const uint32_t* tab = inverse ? t.inv : t.fwd;   // standalone flavour of the inverse table
    if (!inverse) {
        if (layout == LAYOUT_POLY) hipLaunchKernelGGL(ntt_fwd_kernel<LAYOUT_POLY>, grid, 64 * NTT_WPB, 0, s, polys, batch, tab, mapping);
        else hipLaunchKernelGGL(ntt_fwd_kernel<LAYOUT_BRAM>, grid, 64 * NTT_WPB, 0, s, polys, batch, tab, mapping);
    } else {
        if (layout == LAYOUT_POLY) hipLaunchKernelGGL(ntt_inv_kernel<LAYOUT_POLY>, grid, 64 * NTT_WPB, 0, s, polys, batch, tab, mapping);
        else hipLaunchKernelGGL(ntt_inv_kernel<LAYOUT_BRAM>, grid, 64 * NTT_WPB, 0, s, polys, batch, tab, mapping);
    }
    return hipGetLastError();
}

hipError_t launch_ntt_traffic(bool inverse, int32_t* polys, size_t batch, const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    const int grid = grid_for((batch + NTT_WPB - 1) / NTT_WPB, t.num_cus * t.ntt_blocks_per_cu * 4 / NTT_WPB);      // the transforms' own launch shape
    if (inverse) hipLaunchKernelGGL(ntt_traffic_kernel<true>, grid, 64 * NTT_WPB, 0, s, polys, batch);
    else hipLaunchKernelGGL(ntt_traffic_kernel<false>, grid, 64 * NTT_WPB, 0, s, polys, batch);
    return hipGetLastError();
}

hipError_t launch_pointwise(int op, int32_t* c, const int32_t* a, const int32_t* b, const int32_t* acc,
                            size_t batch, const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    const size_t nvec4 = batch * 64;
    const int grid = grid_for((nvec4 + 255) / 256, t.num_cus * 8);
    switch (op) {
    case OP_MUL: hipLaunchKernelGGL(pointwise_kernel<OP_MUL>, grid, 256, 0, s, c, a, b, acc, nvec4); break;
    case OP_MAC: hipLaunchKernelGGL(pointwise_kernel<OP_MAC>, grid, 256, 0, s, c, a, b, acc, nvec4); break;
    case OP_ADD: hipLaunchKernelGGL(pointwise_kernel<OP_ADD>, grid, 256, 0, s, c, a, b, acc, nvec4); break;
    case OP_SUB: hipLaunchKernelGGL(pointwise_kernel<OP_SUB>, grid, 256, 0, s, c, a, b, acc, nvec4); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_polymul(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    const int grid = grid_for((batch + NTT_WPB - 1) / NTT_WPB, t.num_cus * t.ntt_blocks_per_cu * 4 / NTT_WPB);
    hipLaunchKernelGGL(polymul_kernel, grid, 64 * NTT_WPB, 0, s, c, a, b, batch, t.fwd, t.inv_pipe);
    return hipGetLastError();
}

hipError_t launch_bram_mul(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping,
                           const Tables& t, hipStream_t s)
{
    if (batch == 0) return hipSuccess;
    const int grid = grid_for((batch * 64 + 255) / 256, t.num_cus * 8);
    hipLaunchKernelGGL(bram_mul_kernel, grid, 256, 0, s, ram, mul_ram, batch, mapping);
    return hipGetLastError();
}

}  // namespace dil
