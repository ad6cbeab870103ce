// kernels.hpp -- launcher interface between the C-ABI (capi.hip) and the kernels (kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

namespace dil {

// ExpandA switches from two lanes per sponge (latency-bound) to one at this many polynomials: measured crossover at level 3 between
// 1000 keys (49 vs 60 us) and 1500 keys (69 vs 61 us) -- about one wave of sponges per SIMD.  The composite calls' "few keys" paths
// (one launch for ExpandA + its latency-bound neighbours) use the same bound.
constexpr size_t EA_TWO_LANE_MAX = 32768;
extern std::atomic<int> two_lane_max_sponges;      // hash_kernels.hip (option "two_lane_max_sponges")

enum { MAP_NATURAL = 0, MAP_AFTER_NTT = 1, MAP_AFTER_INVNTT = 2 };   // config.h:45-50 (enum MAPPING)
enum { LAYOUT_POLY = 0, LAYOUT_BRAM = 1 };
enum { OP_MUL = 0, OP_MAC = 1, OP_ADD = 2, OP_SUB = 3 };              // butterfly.v modes MULT / ADD / SUB
enum { OUT_W = 0, OUT_W1W0 = 1 };
// Matrix formats in HBM.  A_I32: the public one, [K][L][256] int32 (the reference's bram words).  A_P24: the composite
// calls' internal one for a matrix per item -- coefficients are < 2^23, so they travel as 3 bytes: [K][L][768 bytes], a
// quarter less write traffic for ExpandA and read traffic for the wave-per-item kernels, which are HBM-bound on exactly that stream.
enum { A_I32 = 0, A_P24 = 1 };
// y of the signing loop in HBM.  Y_I32: [L][256] int32 canonical (the public form).  Y_PACKED: the B-bit packed SHAKE256 stream
// ExpandMask squeezes (gamma1 - y; B = 18 | 20, 576 | 640 bytes per polynomial, the wire layout of z) -- internal to dil_sign_*:
// ExpandMask has no rejection, so its packed output IS the stream, and phase 1 / phase 2 unpack in their load stage.
enum { Y_I32 = 0, Y_PACKED = 1 };

struct Tables {
    const uint32_t* fwd = nullptr;   // device, [4][64][8]
    const uint32_t* inv = nullptr;   // device, [4][64][8]  standalone inverse (f = 1/256)
    const uint32_t* inv_pipe = nullptr;   // same, f = 2^32/256: cancels the 2^-32 of the fused pointwise stage
    int device = 0;               // HIP device these tables live on (key of the occupancy cache)
    int num_cus = 256;
    int ntt_blocks_per_cu = 8;    // persistent 256-thread blocks per CU for the NTT kernels   (option ntt_blocks_per_cu)
    int fused_wgs_per_cu = 4;     // persistent workgroups per CU, workgroup-per-item pipelines (option fused_wgs_per_cu)
    int wpi_blocks_per_cu = 8;    // cap on persistent 256-thread blocks per CU (actual = occupancy), wave-per-item (option wpi_blocks_per_cu)
    int fused_mode = 0;           // 0 auto (by batch size), 1 workgroup-per-item, 2 wave-per-item (option fused_mode)
};

hipError_t launch_ntt(bool inverse, int layout, int mapping, int32_t* polys, size_t batch, const Tables& t, hipStream_t s);
// the transforms' loads and stores without the arithmetic (bench.py roofline.achievable); scrambles `polys`
hipError_t launch_ntt_traffic(bool inverse, int32_t* polys, size_t batch, const Tables& t, hipStream_t s);
// ---- host mailbox: the drop-in surface's batch-of-one calls (ntt(), invntt(), pointwise_barrett(), ntt2x2_* on one `bram`) ----
// A launch per call costs ~43 us (two hipMemcpy + a launch + a synchronisation) for 1 KiB of work -- 20 x the CPU reference it
// replaces.  Instead ONE wave stays resident while calls keep coming and serves them from a mailbox in pinned, device-mapped
// host memory: the caller writes its polynomial(s) and bumps `req_seq`; the wave polls that word over PCIe, reads the payload,
// transforms it in registers, writes the result and `done_seq` back; the caller spins on `done_seq` in its own memory.  No
// launch, no hipMemcpy, no stream synchronisation on the path: ~5 us a call (profiles/r04*_mailbox.txt).  The wave retires after
// `idle_ticks` without a request (a resident kernel would otherwise hold hipDeviceSynchronize() forever) and is relaunched by the
// next call; `state` closes the race between "retiring" and "a request just arrived" (capi.hip mailbox_call).
enum { MB_FWD = 0, MB_INV = 1, MB_PW_MUL = 2, MB_BRAM_FWD = 3, MB_BRAM_INV = 4, MB_BRAM_MUL = 5, MB_QUIT = 6, MB_POLYMUL = 7 };
enum { MB_DEAD = 0, MB_ALIVE = 1, MB_EXITING = 2 };
// The request header is ONE 32-bit word -- sequence number << 8 | op << 2 | mapping -- so that the wave's poll can never see a new
// sequence number with the previous request's operation (a 16-byte PCIe read is not guaranteed to be untorn against the host's stores).
__host__ __device__ inline uint32_t mb_header(uint32_t seq, uint32_t op, uint32_t mapping) { return (seq << 8) | (op << 2) | (mapping & 3u); }
struct Mailbox {
    // host -> device (one cache line)
    uint32_t req_seq, pad0[15];           // mb_header(seq, op, mapping)
    // device -> host (one cache line)
    uint32_t done_seq, state, served, pad1[13];
    int32_t in0[256], in1[256], out[256];
};
hipError_t launch_clock_probe(uint64_t* out4, uint64_t spin_ticks, hipStream_t s);   // bench.py: effective shader clock
hipError_t launch_mailbox(Mailbox* mb_dev, uint32_t last_done, uint64_t idle_ticks, uint64_t max_resident_ticks, const Tables& t, hipStream_t s);

hipError_t launch_pointwise(int op, int32_t* c, const int32_t* a, const int32_t* b, const int32_t* acc, size_t batch,
                            const Tables& t, hipStream_t s);
hipError_t launch_bram_mul(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping, const Tables& t, hipStream_t s);
hipError_t launch_polymul(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, const Tables& t, hipStream_t s);   // c = invntt(ntt(a) o ntt(b)), fused
// Optional key indirection of the per-item-key pipelines (the signing loop's speculative entries): entry `it` uses the
// key material of row idx[(base + it) / S] (idx == nullptr: that row number itself).  Default = identity.
constexpr int TICKET_PARTS = 64, TICKET_STRIDE = 32, TICKET_WORDS = TICKET_PARTS * TICKET_STRIDE;
constexpr int FLAG_SUPERSEDED = 16;      // sign-loop flag of an attempt dropped because an earlier attempt of its item was accepted
struct KeyMap {
    const int32_t* idx = nullptr;
    uint32_t S = 1;
    uint32_t base = 0;          // entry number of this launch's item 0 (a round may be split over two streams)
    // Speculative signing rounds (scheme.hip sign_core): entry e = j * S + a is attempt a of pending item j, and only the FIRST
    // accepted attempt of an item is ever used.  spec_n = the round's pending items (0: off) lets the early-exit phase 2 walk the
    // entries attempt-major and drop an entry whose item already shows an accepted earlier attempt (pipelines.hip).
    uint32_t spec_n = 0;
    // Work queues of such a launch (TICKET_WORDS zeroed device words; nullptr: static striding): attempts cost between nothing
    // (dropped) and a full phase 2 (accepted), so the persistent waves draw their next entry from a counter instead of striding.
    // One counter cannot serve a launch (12 ns per atomic on one address: 300 us for 24576 entries, measured): workgroup b draws
    // from queue b % TICKET_PARTS, which holds the entries u = b % TICKET_PARTS (mod TICKET_PARTS), each on its own 128-byte line.
    uint32_t* ticket = nullptr;
    __host__ __device__ size_t key(size_t it) const
    {
        const uint32_t i = (base + (uint32_t)it) / S;
        return idx ? (size_t)idx[i] : (size_t)i;
    }
};
hipError_t launch_matvec(int level, int out_mode, int32_t* w, uint8_t* w1, int32_t* w0, const int32_t* A, const int32_t* y,
                         size_t batch, int shared_A, const Tables& t, hipStream_t s, KeyMap km = KeyMap(),
                         uint8_t* w1_packed = nullptr,    // OUT_W1W0 only: w1 also written packed (4 | 6 bits), [batch][K * 128|192]
                         int a_fmt = A_I32,               // A_P24: a matrix per key in packed form (never with shared_A)
                         int y_fmt = Y_I32);              // Y_PACKED (OUT_W1W0, wave-per-item / shared-key shapes only): y is ExpandMask's raw stream
bool fused_wpi_shape(size_t batch, const Tables& t);      // the wave-per-item / shared-key kernels serve this batch size
// keygen: t = A s1 + s2, Power2Round, t1 -> pk (10 bit), 2^12 - t0 -> sk (13 bit) in one wave-per-key kernel (pipelines.hip)
bool keygen_fused_available(size_t batch, const Tables& t);
hipError_t launch_keygen_matvec(int level, uint8_t* pk, size_t pk_stride, uint8_t* sk, size_t sk_stride, size_t sk_t0_offset,
                                const int32_t* A, const int32_t* s1, const int32_t* s2, size_t batch, const Tables& t, hipStream_t s,
                                int a_fmt = A_I32);
hipError_t launch_verify(int level, uint8_t* w1, const int32_t* A, const int32_t* z, const int32_t* c, const int32_t* t1,
                         const uint8_t* h, size_t batch, int shared_pk, const Tables& t, hipStream_t s);
hipError_t launch_sign2(int level, int32_t* z, uint8_t* h, int32_t* flags, const int32_t* c, const int32_t* y,
                        const int32_t* w0, const uint8_t* w1, const int32_t* s1hat, const int32_t* s2hat,
                        const int32_t* t0hat, size_t batch, int shared_key, const Tables& t, hipStream_t s, KeyMap km = KeyMap(),
                        int32_t* w0_scratch = nullptr,    // == w0: rejected attempts stop at their first failed check
                        int y_fmt = Y_I32,
                        bool small_key = false);          // the caller vouches for |c s1|, |c s2| <= 1023 and |c t0| < 2^18 (a key decoded from sk
                                                          // bytes, c from SampleInBall): SmallPair / exact tails; false: any residues

// ---- the same operations with one sponge per wavefront (coop_kernels.hip): what the launchers below run for calls with few sponges ----
extern std::atomic<int> coop_max_sponges;          // option "coop_max": a call with at most this many sponges runs them one per wavefront (0: never)
bool coop_wanted(size_t sponges);
bool coop_wanted_sib(size_t items);                // SampleInBall's own (1.5 x) bound
hipError_t launch_coop_shake256(uint64_t* out, int out_bytes, const uint64_t* in, int in_bytes, size_t batch, hipStream_t s);
hipError_t launch_coop_challenge_hash(uint8_t* out32, int32_t* verdict, const uint8_t* mu, const uint8_t* w1p, int w1_words, const uint8_t* expect,
                                      size_t expect_stride, size_t batch, hipStream_t s);
hipError_t launch_coop_challenge_sample(uint8_t* ctilde, int32_t* c, const uint8_t* mu, const uint8_t* w1p, int w1_words, int tau, size_t batch, hipStream_t s);
hipError_t launch_coop_sample_in_ball(int32_t* c, uint32_t* cbits, const uint8_t* ctilde, size_t ct_stride, int tau, size_t nitems, hipStream_t s);   // c or cbits
hipError_t launch_coop_expand_mask(void* y, bool raw, const uint8_t* rhoprime, const uint32_t* kappa, int level, size_t nitems, hipStream_t s);
hipError_t launch_coop_expand_a(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, int K, int L, size_t nitems, hipStream_t s);
hipError_t launch_coop_expand_s(int32_t* s1, int32_t* s2, const uint8_t* rhoprime, size_t rp_stride, int eta, int L, int K, size_t nitems, hipStream_t s);
hipError_t launch_coop_expand_a_sib(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, size_t nkeys, int K, int L, uint32_t* cbits, const uint8_t* ctilde,
                                    size_t ct_stride, int tau, size_t nitems, hipStream_t s);
hipError_t launch_coop_expand_a_s(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, int32_t* s1, int32_t* s2, const uint8_t* rhoprime, size_t rp_stride,
                                  int K, int L, int eta, size_t nkeys, hipStream_t s);
hipError_t launch_coop_mu(uint8_t* mu, const uint8_t* tr, size_t tr_stride, const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets,
                          const uint32_t* lengths, int32_t* bad, size_t batch, hipStream_t s);

// ---- row N1: SHAKE-bound samplers (hash_kernels.hip) ----
hipError_t launch_shake256(uint64_t* out, int out_bytes, const uint64_t* in, int in_bytes, size_t batch, hipStream_t s);
hipError_t launch_expand_a(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, int level, size_t nitems, hipStream_t s, int a_fmt = A_I32);
// few keys: A = ExpandA(rho) and (s1, s2) = ExpandS(rho') in one launch, two lanes per sponge (hash_kernels.hip)
hipError_t launch_expand_a_s(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, int32_t* s1, int32_t* s2, const uint8_t* rhoprime,
                             size_t rp_stride, int level, int eta, size_t nkeys, hipStream_t s);
hipError_t launch_expand_mask(int32_t* y, const uint8_t* rhoprime, const uint32_t* kappa, int level, size_t nitems, hipStream_t s);
// y as the raw B-bit stream (Y_PACKED), lane per sponge: the signing loop's large rounds
hipError_t launch_expand_mask_packed(uint8_t* yp, const uint8_t* rhoprime, const uint32_t* kappa, int level, size_t nitems, hipStream_t s);
hipError_t launch_sample_in_ball(int32_t* c, const uint8_t* ctilde, int level, size_t nitems, hipStream_t s);
// the signing loop's challenge in one launch: c~ = H(mu || w1_packed) -> ctilde (32 B per entry) AND c = SampleInBall(c~) -> c (int32 [256] per entry)
hipError_t launch_challenge_sample(uint8_t* ctilde, int32_t* c, const uint8_t* mu, const uint8_t* w1p, int level, size_t batch, hipStream_t s);
hipError_t launch_pack_w1(uint8_t* out, const uint8_t* w1, int level, size_t nitems, const Tables& t, hipStream_t s);
// expect (may be nullptr): 32 bytes per item at expect + i * expect_stride, any alignment (c~ read in place from a signature)
hipError_t launch_challenge_hash(uint8_t* out32, int32_t* verdict, const uint8_t* mu, const uint8_t* w1p, int level,
                                 const uint8_t* expect, size_t batch, hipStream_t s, size_t expect_stride = 32);
// bad (nullable): bad[i] = 1 where (offsets[i], lengths[i]) leaves the blob of msgs_bytes bytes -- such an item is hashed as an empty message
hipError_t launch_mu(uint8_t* mu, const uint8_t* tr, size_t tr_stride, const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets,
                     const uint32_t* lengths, int32_t* bad, size_t batch, hipStream_t s);
// signing of items whose message reference was bad: attempts[i] = -1, signature bytes zero
hipError_t launch_sign_void_bad(uint8_t* sig, size_t sig_bytes, int32_t* attempts, const int32_t* bad, size_t n, hipStream_t s);
hipError_t launch_z_norm(int32_t* verdict, const int32_t* z, int level, size_t batch, hipStream_t s);

// ---- wire-format fused verify (wire_kernels.hip): packed z / t1 / hints / c in, packed w1 + verdict bits 2|4 out ----
hipError_t launch_verify_wire(int level, uint8_t* w1p, int32_t* verdict, const int32_t* A, const uint8_t* pk, size_t pk_stride,
                              const uint8_t* sig, size_t sig_stride, const uint32_t* cbits, size_t batch, int shared_pk,
                              const Tables& t, hipStream_t s, int a_fmt = A_I32,
                              const int32_t* t1hat = nullptr);     // NTT(t1 2^13) per key, kept by the caller (a key per item): the kernel skips those K transforms
hipError_t launch_expand_t1(int32_t* t1hat, const uint8_t* pk, size_t pk_stride, int level, size_t nkeys, const Tables& t, hipStream_t s);
// set-up of a signing call in one launch: [ExpandA of few keys,] s1^ s2^ t0^ = NTT(unpack(sk)), rho' = SHAKE256(key || mu), attempts = 0
hipError_t launch_sign_setup(int level, int32_t* A, bool expand_a_here, int32_t* s1h, int32_t* s2h, int32_t* t0h, const uint8_t* sk,
                             size_t nk, uint8_t* rp, int32_t* attempts, const uint8_t* mu, size_t key_stride, size_t batch, const Tables& t,
                             hipStream_t s);
// ExpandA (two lanes per sponge) of `nkeys` keys and SampleInBall of `nitems` signatures in ONE launch (wire_kernels.hip)
hipError_t launch_expand_a_sib(int32_t* A, const uint8_t* rho, size_t rho_stride_bytes, size_t nkeys, uint32_t* cbits, const uint8_t* ctilde,
                               size_t ct_stride, int level, size_t nitems, hipStream_t s);
hipError_t launch_sample_in_ball_bits(uint32_t* cbits, const uint8_t* ctilde, size_t ct_stride, int level, size_t nitems, hipStream_t s);

// ---- rows N2 / N4: codecs, ExpandS, Power2Round (codec_kernels.hip) ----
enum { XF_PLAIN = 0, XF_OFFSET_MINUS = 1 };
// optional row indirection of the pack-side codecs (the signing loop packs only the accepted attempts):
// work item w reads input row src_row[w], writes output row dst_row[w]; *count (device) bounds w
struct RowMap {
    const int32_t* src_row = nullptr;
    const int32_t* dst_row = nullptr;
    const int32_t* count = nullptr;
};
hipError_t launch_unpack(int bits, int32_t* out, const uint8_t* in, size_t in_stride, size_t in_offset, int polys, int xf,
                         int32_t offset, size_t nitems, const Tables& t, hipStream_t s);
hipError_t launch_pack(int bits, uint8_t* out, size_t out_stride, size_t out_offset, const int32_t* in, int polys, int xf,
                       int32_t offset, size_t nitems, const Tables& t, hipStream_t s, RowMap map = RowMap());
// end of key generation in one launch: tr = H(pk) -> sk, rho / key -> sk, s1 / s2 packed -> sk  (codec_kernels.hip)
hipError_t launch_keygen_finish(uint8_t* sk, size_t sk_bytes, const uint8_t* pk, size_t pk_bytes, const uint8_t* e, const int32_t* s1,
                                const int32_t* s2, int L, int K, int eta, int eta_bits, size_t nkeys, hipStream_t s);
hipError_t launch_hint_unpack(uint8_t* h, int32_t* bad, const uint8_t* in, size_t in_stride, size_t in_offset, int K, int omega,
                              size_t nitems, hipStream_t s);
hipError_t launch_hint_pack(uint8_t* out, size_t out_stride, size_t out_offset, const uint8_t* h, int K, int omega, size_t nitems,
                            hipStream_t s, RowMap map = RowMap());
hipError_t launch_expand_s(int32_t* s1, int32_t* s2, const uint8_t* rhoprime, size_t rp_stride, int eta, int L, int K, size_t nitems,
                           hipStream_t s);
hipError_t launch_power2round(int32_t* t1, int32_t* t0, const int32_t* w, const int32_t* s2, size_t n, const Tables& t, hipStream_t s);
hipError_t launch_or_flag(int32_t* verdict, const int32_t* flag, int bit, size_t n, const Tables& t, hipStream_t s);
hipError_t launch_gather_rows(void* dst, const void* src, const int32_t* idx, size_t row_bytes, uint32_t S, size_t entries,
                              const Tables& t, hipStream_t s);
hipError_t launch_sign_kappa(uint32_t* kappa, int32_t* flags, uint32_t a0, uint32_t L, uint32_t S, size_t entries, hipStream_t s);
hipError_t launch_sign_round_setup(uint8_t* mu_c, uint8_t* rp_c, uint32_t* kappa, int32_t* flags, int32_t* counts, uint32_t* tickets,
                                   const uint8_t* mu, const uint8_t* rp,
                                   const int32_t* idx, uint32_t a0, uint32_t L, uint32_t S, size_t entries, bool gather, hipStream_t s);
hipError_t launch_sign_collect_ct(int32_t* attempts, int32_t* next_idx, int32_t* win_entry, int32_t* win_item, int32_t* counts,
                                  const int32_t* flags, const int32_t* idx, int a0, int S, size_t n, uint8_t* sig, size_t sig_stride,
                                  const uint8_t* ct, hipStream_t s, int32_t* host_words = nullptr, uint32_t seq = 0);
hipError_t launch_copy_field(uint8_t* dst, size_t dst_stride, size_t dst_off, const uint8_t* src, size_t src_stride, size_t src_off,
                             int nbytes, size_t nitems, const Tables& t, hipStream_t s, RowMap map = RowMap());

}  // namespace dil
