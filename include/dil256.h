/*
 * dil256.h -- C-ABI of libdil256.so: the MI355X (gfx950) drop-in for the NTT hot path of
 * GMUCERG/Dilithium's dilithium-256/ C++ code and of the polynomial-ALU sequences of its RTL.
 *
 * Plain C: pointers, sizes, ints.  No torch / C++ types.  Every entry point returns 0 on
 * success or a hipError_t value (see dil_error_string).  Callers own every buffer; the library
 * never retains, frees or reallocates them (the reference's ownership rule: fixed-size,
 * in-place, caller-owned -- ref_ntt.h:30-36, ntt2x2.h:30-34).
 *
 * Data model
 *   polynomial  = int32_t[256] = 1024 B          (data_t[DILITHIUM_N], params.h:30-35)
 *   bram        = int32_t[64][4], row r = coefficients 4r..4r+3   (config.h:29-36, util.cpp:61-72)
 *   batches are dense arrays of polynomials: [batch][256], [batch][L][256], [batch][K][L][256] ...
 *   *_dev entry points take DEVICE pointers + a hipStream_t (as void*, NULL = default stream)
 *   and are asynchronous; *_host entry points take host pointers and are synchronous
 *   (H2D copy, kernel, D2H copy).
 * Value domain
 *   forward transforms (dil_ntt_*, dil_bram_fwdntt_*): any int32 with |x| < 2^31 - 7q (the lazy butterflies
 *   widen a value by < 6q in total and the final reduction needs 2^22 of headroom; tests/test_gpu_ntt.py probes the edge);
 *   inverse transforms and the time-domain inputs of the fused pipelines (y, z, c, w0, t1): |x| < q (canonical
 *   [0, q) or centred); NTT-domain inputs (A, s1hat, s2hat, t0hat, b of pointwise ops) in (-q, q) for the
 *   pointwise ops and canonical [0, q) for the fused pipelines.  ALL outputs are canonical residues in [0, q)
 *   -- the RTL's convention (butterfly.v:194-195); the reference C++ returns (-q, q) and its own tests
 *   compare canonically (util.cpp:98-112, ref_test_ntt_ntt2x2.cpp:31-42).
 * Devices and threads
 *   State is kept PER HIP DEVICE and created on first use by whichever thread has that device current: every
 *   entry point works on the calling thread's current device (hipGetDevice), which must be the device that owns
 *   the buffers and the stream passed in -- HIP's own rule for a kernel launch.  One process can therefore drive
 *   all GPUs of a node: one host thread per GPU (hipSetDevice / dil_init(device) once per thread), or one thread
 *   switching devices between calls.  Calls are thread-safe; concurrent composite calls (dil_keygen / dil_sign /
 *   dil_verify_sig ...) on ONE stream are not ordered against each other's scratch -- use a stream per thread,
 *   as with any stream-ordered API.  The *_host transform entry points of one device serialise on a lock of
 *   their own (they share staging buffers); dil_*_multi_host / dil_*_multi_dev (below) spread a batch over every GPU.
 */
#ifndef DIL256_H
#define DIL256_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DIL_Q 8380417
#define DIL_N 256

/* enum MAPPING of the hardware model (config.h:45-50) */
#define DIL_MAP_NATURAL 0
#define DIL_MAP_AFTER_NTT 1
#define DIL_MAP_AFTER_INVNTT 2

/* ---- life cycle --------------------------------------------------------------------- */
/* dil_init: make `device` the calling thread's current HIP device (-1 = keep the current one) and bring its
 * state up: twiddle tables (derived from zeta = 1753; identical to consts.cpp:64-97 / zetas.txt mod q), a
 * private memory pool.  Optional and idempotent -- every entry point initialises its device on first use.
 * dil_shutdown: free the state of every device this process has used (no call may be in flight). */
int dil_init(int device);
int dil_shutdown(void);
/* Process-wide options (also read once from the environment, names in parentheses):
 *   "fused_mode"  (DIL_FUSED_MODE)  0 = pick the fused-pipeline kernel shape by batch size (default),
 *                                   1 = workgroup-per-item kernels, 2 = wave-per-item / shared-key kernels.
 *                                   Both shapes compute identical results; tests run both on the same input.
 *   "fuse_wire"   (DIL_FUSE_WIRE)   1 = wire-format verification reads packed z / t1 / hints in the fused kernel
 *                                   (default), 0 = separate codec kernels + int32 verify core
 *   "a24"         (DIL_A24)         1 (default) = inside dil_keygen_dev / dil_sign_* a matrix per key crosses HBM as 24-bit
 *                                   packed coefficients (768 bytes per polynomial) when the batch is large: the mat-vec kernels are
 *                                   bound by that stream (level 3, 8192 keys: 85 -> 51 us); 0 = as int32; 2 = packed in
 *                                   dil_verify_sig_dev too (measured neutral there).  Internal only: every A in this header's
 *                                   signatures is int32 [K][L][256]
 *   "fuse_keygen" (DIL_FUSE_KEYGEN) 1 (default) = large key-generation batches run t = A s1 + s2, Power2Round and the packing of
 *                                   t1 into pk / t0 into sk as ONE kernel; 0 = mat-vec, power2round and pack kernels
 *   "fuse_sib"    (DIL_FUSE_SIB)    wire-format verification with a key per signature: c = SampleInBall(c~) is sampled INSIDE the fused
 *                                   verify kernel by the wave that owns the signature (one SHAKE256 state over the wavefront) instead of a
 *                                   sampling launch in front.  bit 0 (default on) = where the matrix is already expanded
 *                                   (dil_verify_wire_core_dev, dil_verify_sig_expanded_dev), bit 1 = in dil_verify_sig_dev too (there the
 *                                   sampler otherwise runs beside ExpandA on the helper stream).  Levels 2 and 3 (level 5: the launch in front is faster and
 *                                   stays).  Verdicts and w1 do not depend on it.
 *   "sign_wake"   (DIL_SIGN_WAKE)   how the pending count of a signing round reaches the host, which sizes the next round from it:
 *                                   1 (default) = the round's last kernel posts it into mapped page-locked words of the library and the
 *                                   calling thread polls them (no copy, no event: the next round is queued while the winners are packed),
 *                                   0 = an 8-byte copy + an event behind that kernel; 2 (tests only) = as 1, but the host waits for a number
 *                                   that is never posted: the wait notices the drained stream and fetches the counts by a blocking copy
 *                                   (the no-hang path).  Signatures and attempt counts do not depend on it.
 *   "zeroize"     (DIL_ZEROIZE)     1 = dil_sign_* / dil_keygen_* clear their device scratch (secret key in NTT
 *                                   form, rho', y, rejected z ...) before returning; 0 (default) = the scratch stays
 *                                   in the per-stream arena until the next call on that stream overwrites it
 *   "sign_skip"   (DIL_SIGN_SKIP)   the signing loop's speculative rounds (several attempts per pending message at once, the first
 *                                   accepted one wins): bit 0 = phase 2 drops an attempt whose message already shows an accepted
 *                                   earlier attempt, bit 1 = its waves draw attempts from work queues instead of striding; default 3.
 *                                   Signatures and attempt counts do not depend on it.
 *   "sign_early", "sign_cap", "sign_waste", "aux_overlap", "ntt_blocks_per_cu", "wpi_blocks_per_cu",
 *   "fused_wgs_per_cu"              tuning knobs
 * Unknown name -> hipErrorInvalidValue.  (Options, device queries and error strings are runtime utilities: the reference -- synthesised
 * logic and a batch-of-one C++ model -- has no counterpart.) */
int dil_set_option(const char* name, int value);
int dil_get_option(const char* name, int* value);
int dil_device_count(int* count);
int dil_num_cus(void);
const char* dil_error_string(int code);
/* Host-only (no GPU needed): the twiddle tables the kernels use, [4 passes][2 halves][64 lanes][4] uint32
 * each (Montgomery form; inv = standalone inverse, inv_pipe = inverse inside the fused
 * pipelines); and the plain 256-entry table (zetas_barrett of consts.h:30, centred). */
void dil_host_twiddle_tables(uint32_t* fwd /*2048*/, uint32_t* inv /*2048*/, uint32_t* inv_pipe /*2048*/);
void dil_host_zetas(int32_t* zetas /*256*/);

/* ---- H2/H3/H5: batched transforms, in place, [batch][256] ------------------------------
 * dil_ntt_*    == ntt() / ntt2x2_ref()        (ref_ntt.cpp:28-47, ref_ntt2x2.cpp:37-82)
 * dil_invntt_* == invntt() / invntt2x2_ref()  (ref_ntt.cpp:59-87, ref_ntt2x2.cpp:100-145) */
int dil_ntt_dev(int32_t* polys, size_t batch, void* stream);
int dil_invntt_dev(int32_t* polys, size_t batch, void* stream);
/* measurement helper (bench.py `roofline.achievable`): the loads and stores of dil_ntt_dev (inverse = 0) / dil_invntt_dev (1) with the
 * same launch shape and NO arithmetic -- what this access pattern reaches on the box.  SCRAMBLES polys: scratch data only. */
int dil_ntt_traffic_dev(int32_t* polys, size_t batch, int inverse, void* stream);
/* Host-pointer forms (`*_host`, here and below): synchronous, caller-owned HOST arrays as in the reference (reference_code/ref_ntt.h:30-36).
 * THE RULE (round 6): a PAGEABLE caller buffer only ever meets memcpy -- its bytes go through the library's own page-locked slots
 * (hipHostMalloc, made once; a ring of three 4-MiB slots on three streams, the memcpy on the calling thread + pool threads, option
 * "host_copy_threads", default 3), so the library never page-locks a page of the caller's memory and never makes the runtime do it: on
 * this platform such locks are per-range attributes shared with every other lock holder of the process (the application's own pageable
 * hipMemcpy / torch copies included), and lock traffic on neighbouring heap ranges killed the round-5 test suite
 * (profiles/r06_suite_crash_rootcause.txt).  A buffer the CALLER page-locked (hipHostMalloc, hipHostRegister, torch pin_memory) is DMA'd
 * in place: one stream per direction from 64 MiB ("host_duplex"), round-robin over "host_streams" streams below; "host_chunk" = KiB per
 * chunk.  Calls of one device serialise on a lock of their own.  dil_host_plan says which form a call takes. */
int dil_ntt_host(int32_t* polys, size_t batch);
int dil_invntt_host(int32_t* polys, size_t batch);

/* ---- H4 + butterfly.v MULT(-ACC)/ADD/SUB modes: element-wise on [batch][256] ------------
 * pointwise: c = a*b (pointwise_barrett, ref_ntt.cpp:49-57; c may alias a or b)
 * mac:       c = acc + a*b (butterfly.v:144-150,224-230; c may alias acc)
 * add / sub: c = a +- b (butterfly.v ADD_MODE / SUB_MODE) */
int dil_pointwise_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream);
int dil_pointwise_acc_dev(int32_t* c, const int32_t* acc, const int32_t* a, const int32_t* b, size_t batch, void* stream);
int dil_poly_add_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream);
int dil_poly_sub_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream);
int dil_pointwise_host(int32_t* c, const int32_t* a, const int32_t* b, size_t batch);

/* ---- the polynomial product chain in one call ------------------------------------------------
 * c = a * b in Z_q[x] / (x^256 + 1) = invntt(pointwise_barrett(ntt(a), ntt(b))): what the reference's callers spell as four calls
 * (hardware_code/ntt2x2_test.cpp:109-137 `polymul`; reference_code/ref_ntt.cpp:28-87), as ONE fused kernel -- both forward transforms, the
 * product and the inverse transform of a pair in one wavefront's registers: 2 KiB read + 1 KiB written per product instead of 9 KiB over
 * four launches.  [batch][256] each, coefficients in natural order, canonical outputs; c may alias a or b; |a|, |b| < 2^26 (the
 * reference's (-q, q) is far inside).  The _host form is the one call that is worth a PCIe round trip for a source-compatible user:
 * a and b go up once, c comes down (batch 1: one mailbox request instead of four). */
int dil_polymul_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream);
int dil_polymul_host(int32_t* c, const int32_t* a, const int32_t* b, size_t batch);

/* ---- H6: the hardware-model API on `bram` ([batch][64][4]) -----------------------------
 * `mapping` is the model's address translation (address_encoder_decoder.cpp:34-55).
 * fwd: NATURAL in -> AFTER_NTT out (ntt2x2_fwdntt.cpp:32-157, ntt2x2_test.cpp:41-58)
 * inv: NATURAL -> AFTER_INVNTT, AFTER_NTT -> NATURAL (ntt2x2_invntt.cpp:38-161, ntt2x2_test.cpp:64-81,129-132)
 * mul: ram[map(l)][k] *= mul_ram[l][k] (ntt2x2_mul.cpp:33-59) */
int dil_bram_fwdntt_dev(int32_t* ram, size_t batch, int mapping, void* stream);
int dil_bram_invntt_dev(int32_t* ram, size_t batch, int mapping, void* stream);
int dil_bram_mul_dev(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping, void* stream);
int dil_bram_fwdntt_host(int32_t* ram, size_t batch, int mapping);
int dil_bram_invntt_host(int32_t* ram, size_t batch, int mapping);
int dil_bram_mul_host(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping);

/* The HOST MAILBOX behind the batch == 1 case of the *_host entry points above (option "host_mailbox" / DIL_HOST_MAILBOX; off by default
 * in libdil256.so, ON in libdil256_ref.so, whose ntt() / invntt() / pointwise_barrett() / ntt2x2_*() are batch-of-one calls by contract).
 * A launch per call costs ~43 us for 1 KiB of work.  With the option on, ONE resident wavefront serves such calls from a mailbox in pinned,
 * device-mapped host memory: the caller's polynomial(s) are memcpy'd in, a sequence word is bumped, the wave -- polling that word over
 * PCIe -- transforms the polynomial in registers and writes result + sequence back; no launch, no hipMemcpy, no synchronisation on the
 * path (~5 us a call).  The wave retires after "mailbox_idle_us" (default 200) without a request, so that hipDeviceSynchronize() is
 * held for at most that long, and the next call relaunches it; a call that finds the mailbox busy (another host thread) or unusable
 * takes the launch path.  Results are bit-identical to the launch path (tests/test_gpu_mailbox.py; the reference's unchanged
 * hardware_code/ntt2x2_test.cpp at its own 10^6 iterations runs through it).  Runtime utility without a reference counterpart.
 * dil_mailbox_stats: requests served through the mailbox / launches of the resident wave so far on the current device, and whether
 * the wave is resident right now (0 retired, 1 serving, 2 retiring). */
int dil_mailbox_stats(uint64_t* calls, uint64_t* launches, int* alive);

/* ---- H9: mat-vec  w[k] = INTT(sum_l A[k][l] o NTT(y[l]))  (combined_top.v:1850-1933) -----
 * level in {2,3,5} selects (K,L) = (4,4)/(6,5)/(8,7).  A: [batch|1][K][L][256] NTT domain,
 * row-major as the RTL stores it (combined_top.v:1370); shared_A != 0 -> one A for the batch.
 * y: [batch][L][256]; w: [batch][K][256]. */
int dil_matvec_dev(int32_t* w, const int32_t* A, const int32_t* y, int level, size_t batch, int shared_A, void* stream);

/* ---- H8: verify core (combined_top.v:1207-1469) -----------------------------------------
 * w1[k] = UseHint(h[k], INTT(sum_l A[k][l] o NTT(z[l]) - NTT(c) o NTT(t1[k] * 2^13)))
 * z: [batch][L][256]; c: [batch][256] (+-1 as 1 / q-1 or -1); t1: [batch|1][K][256] 10-bit,
 * UNscaled (the 2^13 of decoder.v:96-100 is applied on device); h: [batch][K][256] bytes 0/1;
 * w1 out: [batch][K][256] bytes (values < 44 / < 16).  shared_pk != 0 -> one (A, t1). */
int dil_verify_core_dev(uint8_t* w1, const int32_t* A, const int32_t* z, const int32_t* c, const int32_t* t1,
                        const uint8_t* h, int level, size_t batch, int shared_pk, void* stream);

/* The same from HOST arrays (the reference's calling convention for this path is caller-owned host buffers, reference_code/ref_ntt.h:30-36):
 * synchronous; items in chunks, each chunk H2D -> fused kernel -> D2H.  Pageable arrays: a chunk's operands are memcpy'd into one of the
 * library's own page-locked 16-MiB slots (a ring of three, option host_copy_threads) and go up as one copy -- the library never page-locks
 * caller memory nor lets the runtime do it (see the host-pointer rule at dil_ntt_host); arrays the caller page-locked: copies straight from
 * them, round-robin over `host_streams` streams in chunks of host_chunk KiB (at least 64 MiB).  PCIe-bound: 45 KiB up and 1.5 KiB down per
 * level-3 item. */
int dil_verify_core_host(uint8_t* w1, const int32_t* A, const int32_t* z, const int32_t* c, const int32_t* t1, const uint8_t* h, int level,
                         size_t batch, int shared_pk);

/* ---- H10: sign inner loop (combined_top.v:1830-2229) --------------------------------------
 * phase 1 (FSM1 + DECOMP): w = INTT(A o NTT(y)); (w1, w0) = Decompose(w); w0 as residue in [0,q).
 * phase 2 (FSM2): z = y + c*s1, r0 = w0 - c*s2, ct0 = c*t0, h = MakeHint(r0 + ct0, w1);
 *   flags[i]: bit0 ||z|| >= gamma1-beta, bit1 ||r0|| >= gamma2-beta, bit2 ||ct0|| >= gamma2,
 *   bit3 #hints > omega  (norm_check.v:84-105, makehint.v:98-99,176-177); 0 = accept.
 * s1hat [batch|1][L][256], s2hat / t0hat [batch|1][K][256]: NTT domain, canonical.
 * dil_sign_phase2_dev accepts ANY residues in c, s1hat, s2hat, t0hat, w0 (one inverse transform per product, 1 + L + 2 K per attempt,
 * the reference's tests on canonical residues). */
int dil_sign_phase1_dev(uint8_t* w1, int32_t* w0, const int32_t* A, const int32_t* y, int level, size_t batch,
                        int shared_key, void* stream);
int dil_sign_phase2_dev(int32_t* z, uint8_t* h, int32_t* flags, const int32_t* c, const int32_t* y, const int32_t* w0,
                        const uint8_t* w1, const int32_t* s1hat, const int32_t* s2hat, const int32_t* t0hat, int level,
                        size_t batch, int shared_key, void* stream);

/* Phase 2 for a SECRET KEY DECODED FROM KEY BYTES and a challenge from SampleInBall -- what the signing loop (dil_sign_dev) runs.
 * The caller vouches for ||c s1||_inf, ||c s2||_inf <= 1023 and ||c t0||_inf < 2^18: true for every challenge (tau coefficients +-1)
 * with every (s1, s2, t0) a secret-key BYTE STRING can decode to -- eta-bit fields give |s| <= 11 at worst, 13-bit fields t0 in
 * (-2^12, 2^12] -- and false for arbitrary residues, which is why this is an entry point of its own and not a comment on
 * dil_sign_phase2_dev (round-3 advisor finding).  Inside the bound the kernels read c s1[k] and c s2[k] off ONE inverse transform of
 * c^ o (s1^[k] + 2^11 s2^[k]) (1 + 2 K transforms per attempt instead of 1 + L + 2 K) and run the norm checks and MakeHint on exact
 * small integers instead of canonical residues; z, h, flags are the reference's, bit for bit (tests/test_gpu_persistent_parity.py,
 * incl. the extremes of what key bytes decode to).  Outside it the results are undefined (never out-of-bounds accesses).
 * early_exit != 0: the loop's form, see dil_sign_phase2_early_dev below (w0 is then IN/OUT); with shared_key the rows run in turn,
 * r0[k] before z[k] (one transform yields both). */
int dil_sign_phase2_skey_dev(int32_t* z, uint8_t* h, int32_t* flags, const int32_t* c, const int32_t* y, int32_t* w0,
                             const uint8_t* w1, const int32_t* s1hat, const int32_t* s2hat, const int32_t* t0hat, int level,
                             size_t batch, int shared_key, int early_exit, void* stream);

/* Phase 2 with the signing loop's EARLY EXIT, for arbitrary residues (dil_sign_dev): an attempt is abandoned at its FIRST failed check; flags != 0 iff the attempt is
 * rejected and names that first check: 2 an r0 row, 1 a z row, 4 a c t0 row (| 8 for too many hints counted so far; 8 alone: all checks
 * ran, hint count over omega).  Order of evaluation: all r0 rows, then all z rows, then the c t0 rows (dil_sign_phase2_skey_dev with
 * shared_key: rows k = 0 .. K-1 in turn, r0[k] before z[k]).  z and h are complete only where flags == 0.
 * w0 is IN/OUT: on return it holds r0 = w0 - c s2 of the rows that were evaluated.  (The reference's FSM2, combined_top.v:1981-2229,
 * evaluates every check and tests `reject_mh || norm_rejected` at the end, :2218; stopping early is this runtime's.)  Batches below
 * the wave-per-item threshold (8 x #CUs items, option fused_mode) run the full phase 2 instead: every flag bit, w0 untouched. */
int dil_sign_phase2_early_dev(int32_t* z, uint8_t* h, int32_t* flags, const int32_t* c, const int32_t* y, int32_t* w0,
                              const uint8_t* w1, const int32_t* s1hat, const int32_t* s2hat, const int32_t* t0hat, int level,
                              size_t batch, int shared_key, void* stream);

/* Debug record of the most recent launch of a persistent kernel family ("verify_wpi", "sign2_wpi", "matvec_wpi", "sign1_wpi",
 * "keygen_wpi", "sign2_early_wpi", "verify_wire_wpi", "matvec_shared", "sign1_shared", "verify_shared", "verify_wire_shared" ...):
 * grid (workgroups), items per workgroup and step, batch, launches so far.  A wave re-enters its item loop when
 * items > grid * items_per_block; the parity tests assert exactly that.  Runtime utility without a reference counterpart. */
int dil_launch_info(const char* family, int* grid, int* items_per_block, size_t* items, size_t* launches);

/* What a host-pointer transform call (dil_ntt_host, dil_invntt_host, dil_bram_*_host) of `batch` polynomials would do under the current
 * options, for a pageable (0) or caller-page-locked (1) buffer: *pipeline = 0 one upload / launch / download, 1 chunks round-robin over the
 * streams (page-locked), 2 one stream per direction (page-locked buffers from 64 MiB), 3 slices through the ring of the library's own
 * page-locked slots (pageable buffers above 4 MiB); *chunk_polys = polynomials per chunk.  Pure host logic (no device needed); the reference's calling convention is
 * caller-owned host arrays (reference_code/ref_ntt.h:30-36).  Runtime utility without a reference counterpart. */
int dil_host_plan(size_t batch, int page_locked, int* pipeline, size_t* chunk_polys);

/* The signing loop's speculation rule as a pure function (no device needed): a call of `batch` messages at `level` with `pending`
 * messages still unsigned after `attempts_done` attempts each runs *attempts_per_item speculative attempts (kappa = attempts_done * L,
 * ...) for every pending message in its next round = *entries entries in flight (options sign_cap, sign_waste).  The reference's FSM
 * retries one signature at a time (rtl_src/combined_top.v:1694-2229); the first accepted attempt of an item is the one it would produce. */
int dil_sign_round_plan(int level, size_t batch, size_t pending, int attempts_done, int max_attempts, int* attempts_per_item, size_t* entries);

/* ---- SHAKE-bound samplers on the device ---------------------------------
 * (round-3 v3.1 conventions, the ones the reference's KAT files obey; all buffers 8-byte aligned)
 * shake256:        out[i] = SHAKE256(in[i]); one input length for the batch; in_bytes, out_bytes % 8 == 0
 * expand_a:        A[i][k][l] from rho[i] (32 B)          gen_a_ext.v, sampler_a_ext.v:129, rejection_a.v:67-73
 * expand_mask:     y[i][l] from rho'[i] (64 B), nonce kappa[i] + l; canonical [0,q)
 *                                                          expandmask_ext.v:98, sampler_y_ext.v, rejection_y.v
 * sample_in_ball:  c[i] (+-1 as 1 / q-1) from c~[i] (32 B) gen_c.v:163-196,318-339
 * pack_w1:         [K][256] bytes -> 4-bit (levels 3/5) or 6-bit (level 2) stream   encoder.v:96-133
 * challenge:       the signing loop's challenge as the ONE unit gen_c.v is: c~[i] = SHAKE256(mu[i] (64 B) || w1_packed[i]
 *                  (K * 128 B; level 2: K * 192 B), 32) -> ctilde, and c[i] = SampleInBall(c~[i]) -> c, in one launch with
 *                  c~ never leaving the sponge's registers in between               gen_c.v:163-196 (absorb), :318-339 (sample) */
int dil_shake256_dev(uint8_t* out, size_t out_bytes, const uint8_t* in, size_t in_bytes, size_t batch, void* stream);
int dil_expand_a_dev(int32_t* A, const uint8_t* rho, int level, size_t batch, void* stream);
int dil_expand_mask_dev(int32_t* y, const uint8_t* rhoprime, const uint32_t* kappa, int level, size_t batch, void* stream);
int dil_sample_in_ball_dev(int32_t* c, const uint8_t* ctilde, int level, size_t batch, void* stream);
int dil_challenge_dev(uint8_t* ctilde, int32_t* c, const uint8_t* mu, const uint8_t* w1_packed, int level, size_t batch, void* stream);
int dil_pack_w1_dev(uint8_t* out, const uint8_t* w1, int level, size_t batch, void* stream);

/* ---- wire-format codecs, ExpandS, keygen ------------------------------
 * Wire formats are those of the reference's KAT files (round-3 v3.1): pk = rho(32) | t1 (K x 320 B);
 * sk = rho(32) | key(32) | tr(32) | s1 (L x 96|128 B) | s2 (K x 96|128) | t0 (K x 416);
 * sig = c~(32) | z (L x 576|640 B) | hint (omega + K B).   Buffers: any alignment, `stride` bytes per item.
 * kind: DIL_CODEC_*; unpack -> int32 [batch][polys][256] canonical; pack <- any residues. */
#define DIL_CODEC_T1 0   /* 10 b, K polys                          decoder.v:96-100 (without the 2^13) */
#define DIL_CODEC_T0 1   /* 13 b, K polys, 2^12 - t0               uncenter_coeff.v:49-65 */
#define DIL_CODEC_S1 2   /* 3|4 b, L polys, eta - s */
#define DIL_CODEC_S2 3   /* 3|4 b, K polys, eta - s */
#define DIL_CODEC_Z 4    /* 18|20 b, L polys, gamma1 - z */
int dil_unpack_dev(int32_t* out, const uint8_t* in, size_t in_stride, size_t in_offset, int kind, int level, size_t batch, void* stream);
int dil_pack_dev(uint8_t* out, size_t out_stride, size_t out_offset, const int32_t* in, int kind, int level, size_t batch, void* stream);
/* hints: omega position bytes + K cumulative counts <-> h [batch][K][256] bytes 0/1; bad[i] = 1 if malformed
 * (decoder side usehint.v:92-114, encoder side makehint.v:104-150) */
int dil_hint_unpack_dev(uint8_t* h, int32_t* bad, const uint8_t* in, size_t in_stride, size_t in_offset, int level, size_t batch, void* stream);
int dil_hint_pack_dev(uint8_t* out, size_t out_stride, size_t out_offset, const uint8_t* h, int level, size_t batch, void* stream);
/* ExpandS: s1 [batch][L][256], s2 [batch][K][256] canonical from rho' (64 B at rhoprime + i*stride)   gen_s.v */
int dil_expand_s_dev(int32_t* s1, int32_t* s2, const uint8_t* rhoprime, size_t stride, int level, size_t batch, void* stream);
/* key generation, seed (32 B) -> pk, sk in wire format (combined_top.v KG_* :754-1079), all on the device */
int dil_keygen_dev(uint8_t* pk, uint8_t* sk, const uint8_t* seed, int level, size_t batch, void* stream);
size_t dil_pk_bytes(int level);
size_t dil_sk_bytes(int level);
size_t dil_sig_bytes(int level);
/* wire-format verification (combined_top.v verify mode: VY_* states :1104-1470; stream order of pk / sig in rtl_tb/tb_verify_top.v:58-68):
 * verdict[i] = 0 accept; bit0 challenge mismatch, bit1 ||z|| bound (norm_check.v:84-105), bit2 malformed hint (usehint.v:92-114).
 * mu [batch][64] = SHAKE256(tr || message) (dil_mu_dev / dil_verify_msg_dev hash the message on the device); shared_pk: one pk for all */
int dil_verify_sig_dev(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level, size_t batch,
                       int shared_pk, void* stream);

/* dil_verify_sig_dev against keys whose matrix A = ExpandA(rho) the caller expanded ONCE (dil_expand_a_dev on the keys' rho,
 * [batch|1][K][L][256] int32, 16-byte aligned) and keeps across calls: many signatures under few public keys.  pk is still
 * read for t1.  Same verdict bits.  (ExpandA is 2/3 of a distinct-key verification batch and 1/3 of a one-key batch.) */
/* ... and with t1^ = NTT(t1 2^13) of every key kept too (rtl_src/combined_top.v:1259-1313 VY_NTT_T1, decoder.v:96-100, once per key):
 * t1hat [nkeys][K][256] int32 canonical from dil_expand_t1_dev; the fused kernel skips those K forward transforms (a key per signature;
 * with shared_pk the batch's one t1^ lives in LDS anyway and t1hat is not read). */
int dil_expand_t1_dev(int32_t* t1hat, const uint8_t* pk /* [nkeys][pk_bytes] */, int level, size_t nkeys, void* stream);
int dil_verify_sig_expanded2_dev(int32_t* verdict, const int32_t* A, const int32_t* t1hat, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu,
                                 int level, size_t batch, int shared_pk, void* stream);
int dil_verify_sig_expanded_dev(int32_t* verdict, const int32_t* A, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level,
                                size_t batch, int shared_pk, void* stream);

/* The fused kernel inside dil_verify_sig_dev (wire_kernels.hip), exposed for parity tests and profiling: reads pk
 * ([batch|1][pk_bytes]) and sig ([batch][sig_bytes]) in wire format -- packed z, t1, hints; c = SampleInBall(c~) -- with
 * A [batch|1][K][L][256] already expanded, and writes w1 PACKED ([batch][K * 128|192] bytes, encoder.v:96-133) plus
 * verdict[i] = bit1 (value 2) ||z|| >= gamma1 - beta | bit2 (value 4) malformed hint encoding.
 * A == NULL: hipErrorInvalidValue.  (Round 2's variant that sampled A = ExpandA(rho) inside the verifying kernel measured slower
 * than ExpandA -> HBM -> this kernel, profiles/r02_gen_a.txt, and was removed in round 4.) */
int dil_verify_wire_core_dev(uint8_t* w1_packed, int32_t* verdict, const int32_t* A, const uint8_t* pk, const uint8_t* sig, int level,
                             size_t batch, int shared_pk, void* stream);

/* The whole deterministic signing loop (combined_top.v sign FSMs :1694-2229) for a batch: sk wire format
 * ([batch][sk_bytes], or one key if shared_sk), mu [batch][64] -> sig [batch][sig_bytes], attempts[i] = number of
 * ATTEMPTS (values of kappa / L tried, the accepted one included) item i took -- the count the sequential reference
 * loop would report; 0 = not finished within max_attempts -> return DIL_ERR_UNFINISHED.
 * Pending items are re-tried together in wide speculative rounds (several attempts per item per round, the first
 * accepted one wins); synchronises `stream` once per ROUND (about 5 rounds for a large batch). */
#define DIL_ERR_UNFINISHED (-2)
int dil_sign_dev(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* mu, int level, size_t batch, int shared_sk,
                 int max_attempts, void* stream);

/* ---- messages in, not digests: mu = SHAKE256(tr || M, 64) on the device --------------------------------------------------
 * The reference's top level absorbs (mlen, tr, m) itself (rtl_src/expandmask_ext.v:131-185; bus order rtl_tb/tb_sign_top.v:57-69,
 * tb_verify_top.v:58-68; its KAT messages are 33 ... 3300 bytes).  Messages are RAGGED: one byte blob `msgs` of `msgs_bytes` bytes plus,
 * per item, offsets[i] (uint64, byte offset into the blob) and lengths[i] (uint32); any alignment, zero length allowed; msgs may be
 * NULL when msgs_bytes == 0 (every message empty).  An item whose (offset, length) leaves the blob is never read past the blob: it is
 * hashed as an empty message and FLAGGED -- dil_mu_dev: bad[i] = 1 (bad may be NULL); dil_sign_msg_dev: attempts[i] = -1 and a zeroed
 * signature (`attempts` is therefore MANDATORY there: NULL -> hipErrorInvalidValue); dil_verify_msg_dev: verdict bit3 (value 8).
 * dil_mu_dev:         mu[i] (64 B, 8-byte aligned) from tr at tr + i * tr_stride (32 B, 8-byte aligned; stride 0 = one tr)
 * dil_sign_msg_dev:   dil_sign_dev on (sk, M): tr is read from the secret key
 * dil_verify_msg_dev: dil_verify_sig_dev on (pk, M, sig): tr = SHAKE256(pk) is computed on the device first */
int dil_mu_dev(uint8_t* mu, const uint8_t* tr, size_t tr_stride, const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets,
               const uint32_t* lengths, int32_t* bad, size_t batch, void* stream);
int dil_sign_msg_dev(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets,
                     const uint32_t* lengths, int level, size_t batch, int shared_sk, int max_attempts, void* stream);
int dil_verify_msg_dev(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets,
                       const uint32_t* lengths, int level, size_t batch, int shared_pk, void* stream);

/* host-buffer forms of the three whole operations (what the reference's test benches tb_keygen_top.v / tb_sign_top.v /
 * tb_verify_top.v stream through the 64-bit port): H2D, the device call, D2H; synchronous */
int dil_keygen_host(uint8_t* pk, uint8_t* sk, const uint8_t* seed, int level, size_t batch);
int dil_sign_host(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* mu, int level, size_t batch, int shared_sk,
                  int max_attempts);
int dil_verify_sig_host(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level, size_t batch,
                        int shared_pk);

/* ---- all GPUs of the node from one C++ process: contiguous slices [g*B/G, (g+1)*B/G) of a batch, one host thread per
 * device running the single-device entry point on its slice; ndev <= 0 = every visible device.  dil_shard_range gives the slice of
 * `rank` (sizes differ by at most one item; same rule as dilithium_amd/sharding.py).
 * HOST buffers (dil_*_multi_host): every thread's D2H copy lands in the caller's array -- no collective. */
void dil_shard_range(size_t n_items, int rank, int world, size_t* lo, size_t* hi);
int dil_ntt_multi_host(int32_t* polys, size_t batch, int inverse, int ndev);
int dil_keygen_multi_host(uint8_t* pk, uint8_t* sk, const uint8_t* seed, int level, size_t batch, int ndev);
int dil_sign_multi_host(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* mu, int level, size_t batch, int shared_sk,
                        int max_attempts, int ndev);
int dil_verify_sig_multi_host(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level, size_t batch,
                              int shared_pk, int ndev);
/* DEVICE-resident data (dil_*_multi_dev): the ONE exchange of the design is an RCCL collective over xGMI of the result slabs.
 *   inputs  in[g]:  device g's SLICE, items [lo_g, hi_g) of dil_shard_range(batch, g, ndev), in device g's memory (one key for the
 *                   batch: a copy per device)
 *   outputs out[g]: a FULL-size array [batch][...] in device g's memory; device g writes its slab in place at item lo_g, then
 *                   gather_root < 0: all-gather -- every out[g] complete (ncclAllGather; ragged slices: one ncclBroadcast per slab in
 *                   one group);  gather_root = r: only out[r] complete (grouped ncclSend / ncclRecv)
 * dil_multi_init builds the communicators (ncclCommInitAll over devices 0 .. ndev-1) and one stream per device, cached until ndev
 * changes or dil_multi_shutdown; the calls below do it on first use.  They are synchronous (every device's stream is drained) and
 * not re-entrant.  RCCL is bound with dlopen at first use: failures return DIL_ERR_RCCL (text: dil_multi_last_error). */
#define DIL_ERR_RCCL (-3)
int dil_multi_init(int ndev);
int dil_multi_shutdown(void);
const char* dil_multi_last_error(void);
/* which RCCL the collectives are bound to: NCCL_VERSION_CODE (major * 10000 + minor * 100 + patch), the library's path, devices of the live communicators (0: none) */
int dil_multi_info(int* rccl_version, char* path, size_t path_len, int* ndev);
/* The schedule of the final gather as pure host logic (no device, no RCCL needed: testable anywhere): the RCCL calls the *_multi_dev entry
 * points issue inside ONE ncclGroupStart / ncclGroupEnd, in issue order, for `batch` items of `item_bytes` over ndev devices.
 *   kind DIL_GATHER_ALLGATHER  device `rank` contributes [offset, offset + bytes) of its own full-size array, in place (equal slabs)
 *        DIL_GATHER_BROADCAST  device `rank` takes part in the broadcast of [offset, offset + bytes) from device `peer` (ragged slabs)
 *        DIL_GATHER_RECV       device `rank` (the root) receives [offset, offset + bytes) from device `peer`
 *        DIL_GATHER_SEND       device `rank` sends its own slab [offset, offset + bytes) to device `peer` (the root)
 * gather_root < 0: every device ends up with every item; gather_root = r: device r alone.  force_ragged != 0: the broadcast form even when
 * the slabs are equal (what option multi_group_at_1 = 2 runs on one device).  Writes at most max_ops records, *n_ops = how many the
 * schedule has.  There is no reference counterpart (the FPGA is one stream port, rtl_src/combined_top.v:36-41); contract: SURVEY 8(e). */
enum { DIL_GATHER_ALLGATHER = 0, DIL_GATHER_BROADCAST = 1, DIL_GATHER_RECV = 2, DIL_GATHER_SEND = 3 };
typedef struct dil_gather_op {
    int kind, rank, peer;
    size_t offset, bytes;
} dil_gather_op;
int dil_multi_gather_plan(size_t batch, size_t item_bytes, int gather_root, int ndev, int force_ragged, dil_gather_op* ops, size_t max_ops, size_t* n_ops);
int dil_gather_slabs_multi_dev(void* const* bufs, size_t item_bytes, size_t batch, int gather_root, int ndev);
int dil_ntt_multi_dev(int32_t* const* polys /* in/out: full-size, slab in place */, size_t batch, int inverse, int gather_root, int ndev);
int dil_sign_multi_dev(uint8_t* const* sig, int32_t* const* attempts /* or NULL */, const uint8_t* const* sk, const uint8_t* const* mu,
                       int level, size_t batch, int shared_sk, int max_attempts, int gather_root, int ndev);
int dil_verify_sig_multi_dev(int32_t* const* verdict, const uint8_t* const* pk, const uint8_t* const* sig, const uint8_t* const* mu, int level,
                             size_t batch, int shared_pk, int gather_root, int ndev);
/* BASELINE configs[4]: sign inner loop (phase 1 + phase 2, combined_top.v:1830-2229) on every device's slice of the attempts, then the
 * gather of the (z, h, flags) slabs.  w1_scratch / w0_scratch [g]: [slice][K][256] bytes / int32 on device g. */
int dil_sign_phases_multi_dev(int32_t* const* z, uint8_t* const* h, int32_t* const* flags, const int32_t* const* A, const int32_t* const* y,
                              const int32_t* const* c, const int32_t* const* s1hat, const int32_t* const* s2hat, const int32_t* const* t0hat,
                              uint8_t* const* w1_scratch, int32_t* const* w0_scratch, int level, size_t batch, int shared_key, int gather_root,
                              int ndev);

/* ---- SURVEY 8(f) row N3 (first step): whole verify / sign-attempt sequences as ONE call --------
 * Everything between the wire-format codecs runs on the device, on `stream`, with no host round trip;
 * temporaries come from the stream-ordered allocator (hipMallocAsync) and are freed on the stream.
 *
 * dil_verify_dev   (combined_top.v VY_* :1149-1534):  c = SampleInBall(c~);  w1 = verify core;
 *                  c~' = SHAKE256(mu || pack(w1));  verdict[i] = 0 accept, bit0 c~' != c~, bit1 ||z|| >= gamma1-beta
 *                  ctilde [B][32], mu [B][64] bytes; z, t1, h, A as for dil_verify_core_dev.
 * dil_sign_attempt_dev (FSM1 + FSM2, :1830-2229): y = ExpandMask(rho', kappa);  (w1,w0) = phase 1;
 *                  c~ = SHAKE256(mu || pack(w1));  c = SampleInBall(c~);  (z,h,flags) = phase 2.
 *                  rhoprime [B][64], kappa [B] (nonce base of this attempt), outputs ctilde [B][32],
 *                  z [B][L][256], h [B][K][256] bytes, flags [B] (0 = accept; the caller re-submits the
 *                  rejected items with kappa += L). */
int dil_verify_dev(int32_t* verdict, const int32_t* A, const uint8_t* ctilde, const int32_t* z, const int32_t* t1,
                   const uint8_t* h, const uint8_t* mu, int level, size_t batch, int shared_pk, void* stream);
int dil_sign_attempt_dev(uint8_t* ctilde, int32_t* z, uint8_t* h, int32_t* flags, const int32_t* A, const uint8_t* mu,
                         const uint8_t* rhoprime, const uint32_t* kappa, const int32_t* s1hat, const int32_t* s2hat,
                         const int32_t* t0hat, int level, size_t batch, int shared_key, void* stream);

/* Effective shader clock over an interval (bench.py): a one-lane kernel on `stream` samples the shader cycle counter and the constant
 * 100 MHz counter, dozes spin_us microseconds and samples again: out4 (device, 4 x uint64) = {cycles0, cycles1, ticks0, ticks1};
 * clock [MHz] = (cycles1 - cycles0) / (ticks1 - ticks0) x 100.  Launch it on a side stream around the kernels being timed. */
int dil_clock_probe_dev(uint64_t* out4, unsigned spin_us, void* stream);

/* ---- timing helpers (hipEvent on the caller's stream; used by bench.py; runtime utilities without a reference counterpart) ---- */
int dil_event_create(void** ev);
int dil_event_destroy(void* ev);
int dil_event_record(void* ev, void* stream);
int dil_event_elapsed_ms(float* ms, void* start, void* stop);   /* synchronises on `stop` */
int dil_stream_sync(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIL256_H */
