// scheme.hip -- SURVEY 8(f) rows N1-N4 behind include/dil256.h: the composite entry points that run whole Dilithium
// operations on the device (single-call verify / sign attempt, wire-format codecs, key generation, verification and
// the signing rejection loop from bytes, and their host-buffer forms).  The kernels are in hash_kernels.hip,
// codec_kernels.hip and pipelines.hip; this file only sequences them on streams.
#include "capi_internal.hpp"

#include <algorithm>
#include <stdlib.h>

using dil::rt::S;
using dil::rt::Arena;
using dil::rt::ArenaPool;
using dil::rt::Device;

// (every dil_* function below is declared extern "C" in include/dil256.h and inherits that linkage)

// ---- row N3 (first step): composite sequences, device-resident end to end -----------------------
// Temporaries of a composite call.  Each stream that makes composite calls gets a grow-only device arena (plus two
// pinned host words) that is reused from call to call: calls on one stream are ordered, so the next call may overwrite
// what the previous one used.  A call carves its buffers out of the arena; whatever does not fit (first call, or a
// bigger batch than ever before) comes from the device's private stream-ordered pool for this call only, and the arena
// is regrown to the new high-water mark afterwards.  Steady state: no allocation and no free per call (25 hipFreeAsync
// calls were costing a signing call 1 ms of host time).  Arenas are per DEVICE (capi_internal.hpp).
Arena* ArenaPool::acquire(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(mu);
    Arena *free_slot = nullptr, *lru = nullptr;
    for (Arena& a : slots) {
        if (a.in_use) {
            if (a.stream == s) return nullptr;      // this stream's arena is busy (or being regrown): the call spills
            continue;
        }
        if (a.base && a.stream == s) {
            a.in_use = true;
            a.last_use = ++tick;
            return &a;
        }
        if (!a.base && !free_slot) free_slot = &a;
        if (a.base && (!lru || a.last_use < lru->last_use)) lru = &a;
    }
    if (!free_slot && lru) {      // every slot belongs to some other (possibly long gone) stream: evict the stalest
        (void)hipFree(lru->base);
        if (lru->pinned) (void)hipHostFree(lru->pinned);
        *lru = Arena();
        free_slot = lru;
    }
    if (!free_slot) return nullptr;
    free_slot->stream = s;
    free_slot->in_use = true;
    free_slot->last_use = ++tick;
    return free_slot;
}
int ArenaPool::release(Arena* a, size_t wanted)
{
    // The slot is ours alone while in_use is set (acquire() skips it), but other threads READ every slot's base / stream under
    // `mu`: the regrow happens on locals and base / size are published together with in_use = false under the mutex.
    // (hipFree waits for the work that still uses the old block; it runs outside the lock so that it stalls nobody else's acquire.)
    int rc = 0;
    char* base = a->base;
    size_t size = a->size;
    const bool regrow = wanted > size;
    if (regrow) {
        {
            std::lock_guard<std::mutex> lk(mu);
            a->base = nullptr;        // nobody may match this slot by stream while its block is being replaced
            a->size = 0;
        }
        if (base) (void)hipFree(base);
        base = nullptr;
        size = 0;
        const size_t sz = wanted + wanted / 8;
        const hipError_t e = hipMalloc(reinterpret_cast<void**>(&base), sz);
        if (e == hipSuccess) {
            size = sz;
        } else {                  // reported to the caller (StreamScratch::close): the next call would spill everything
            (void)hipGetLastError();
            base = nullptr;
            rc = (int)e;
        }
    }
    std::lock_guard<std::mutex> lk(mu);
    if (regrow) {
        a->base = base;
        a->size = size;
    }
    a->in_use = false;
    return rc;
}
void ArenaPool::clear()
{
    std::lock_guard<std::mutex> lk(mu);
    for (Arena& a : slots) {
        if (a.in_use) continue;
        if (a.base) (void)hipFree(a.base);
        if (a.pinned) (void)hipHostFree(a.pinned);
        a = Arena();
    }
}

namespace {

struct StreamScratch {
    Device& dv;
    hipStream_t s;
    Arena* arena;
    size_t used = 0;             // bytes carved or wanted so far (256-byte granules)
    void* spill[40];             // buffers that did not fit: stream-ordered pool, this call only
    size_t spill_bytes[40];
    int nspill = 0;
    int rc = 0;                  // first allocation failure; check once after the last take()
    bool closed = false;
    bool secret = false;         // the call puts key-dependent data here: wiped on close when option `zeroize` is set
    StreamScratch(Device& d, hipStream_t st) : dv(d), s(st), arena(d.arenas.acquire(st)) {}
    // arena looked up under `key` (any unique handle), spills allocated on `st`
    StreamScratch(Device& d, hipStream_t key, hipStream_t st) : dv(d), s(st), arena(d.arenas.acquire(key)) {}
    template <class T>
    T* take(size_t count)
    {
        const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        const size_t off = used;
        used += bytes ? bytes : 256;
        if (arena && off + bytes <= arena->size) return reinterpret_cast<T*>(arena->base + off);
        void* q = nullptr;
        if (rc == 0 && nspill < 40) {
            const size_t nb = bytes ? bytes : 256;
            hipError_t e = dv.pool ? hipMallocFromPoolAsync(&q, nb, dv.pool, s) : hipMallocAsync(&q, nb, s);
            if (e != hipSuccess) {
                rc = (int)e;
            } else {
                spill_bytes[nspill] = nb;
                spill[nspill++] = q;
            }
        } else if (rc == 0) {
            rc = (int)hipErrorOutOfMemory;
        }
        return static_cast<T*>(q);
    }
    // four page-locked host words, mapped and coherent (per arena; a private allocation when the call has no arena): the signing loop's
    // [0] pending, [1] winners, [2] the sequence number the device posts behind them, [3] the host's own sequence counter
    int32_t* own_pinned = nullptr;
    int32_t* pinned_words()
    {
        int32_t** slot = arena ? &arena->pinned : &own_pinned;
        if (!*slot) {
            if (hipHostMalloc(reinterpret_cast<void**>(slot), 4 * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
                *slot = nullptr;
                rc = rc ? rc : (int)hipErrorOutOfMemory;
            } else {
                for (int i = 0; i < 4; i++) (*slot)[i] = 0;
            }
        }
        return *slot;
    }
    // End of the call: wipe (option `zeroize`, secret-bearing calls only), hand spills back, regrow the arena to the
    // high-water mark.  Returns `status`, or -- when that is 0 -- the error of a failed wipe / regrow, so an arena
    // that could not be regrown is reported instead of silently degrading to per-call spills.
    int close(int status)
    {
        if (closed) return status;
        closed = true;
        int err = 0;
        if (secret && dil::rt::cfg.zeroize.load(std::memory_order_relaxed)) {
            if (arena && arena->base && used) {
                const hipError_t e = hipMemsetAsync(arena->base, 0, used < arena->size ? used : arena->size, s);
                if (e != hipSuccess && !err) err = (int)e;
            }
            for (int i = 0; i < nspill; i++) {
                const hipError_t e = hipMemsetAsync(spill[i], 0, spill_bytes[i], s);
                if (e != hipSuccess && !err) err = (int)e;
            }
        }
        for (int i = 0; i < nspill; i++) (void)hipFreeAsync(spill[i], s);
        if (own_pinned) (void)hipHostFree(own_pinned);
        if (arena) {
            const int e = dv.arenas.release(arena, used);
            if (e && !err) err = e;
        }
        return status ? status : err;
    }
    ~StreamScratch() { (void)close(0); }
};
int level_kl(int level, int* K, int* L)
{
    switch (level) {
    case 2: *K = 4; *L = 4; return 0;
    case 3: *K = 6; *L = 5; return 0;
    case 5: *K = 8; *L = 7; return 0;
    default: return (int)hipErrorInvalidValue;
    }
}
}  // namespace

int dil_verify_dev(int32_t* verdict, const int32_t* A, const uint8_t* ctilde, const int32_t* z, const int32_t* t1,
                   const uint8_t* h, const uint8_t* mu, int level, size_t batch, int shared_pk, void* stream)
{
    int rc, K, L;
    if ((rc = level_kl(level, &K, &L))) return rc;
    DIL_ENTER(dv, T);
    if (batch == 0) return 0;
    hipStream_t s = S(stream);
    StreamScratch ws(dv, s);
    int32_t* c = ws.take<int32_t>(batch * 256);
    uint8_t* w1 = ws.take<uint8_t>(batch * K * 256);
    uint8_t* w1p = ws.take<uint8_t>(batch * K * (level == 2 ? 192 : 128));
    if (ws.rc) return ws.rc;
    DIL_TRY(dil::launch_z_norm(verdict, z, level, batch, s));
    DIL_TRY(dil::launch_sample_in_ball(c, ctilde, level, batch, s));
    DIL_TRY(dil::launch_verify(level, w1, A, z, c, t1, h, batch, shared_pk, T, s));
    DIL_TRY(dil::launch_pack_w1(w1p, w1, level, batch, T, s));
    DIL_TRY(dil::launch_challenge_hash(nullptr, verdict, mu, w1p, level, ctilde, batch, s));
    return ws.close(0);
}

namespace {
// temporaries of one signing attempt over `batch` entries
struct AttemptScratch {
    int32_t* y; uint8_t* w1; int32_t* w0; uint8_t* w1p; int32_t* c;
    int a_fmt = dil::A_I32;      // format of the matrix the attempts multiply by (kernels.hpp)
    bool fuse_challenge = true;  // c~ and c by one launch (hash_kernels.hip challenge_sample_kernel)
    bool packed_y = false;       // the signing loop: y stays ExpandMask's raw stream in rounds large enough for the wave-per-item kernels
    bool small_key = false;      // s1^ s2^ t0^ decoded from secret-key bytes, c from SampleInBall: phase 2 may use its small-product forms
    bool w0w1 = false;           // the signing loop: between phase 1 and phase 2 w0 and w1 travel as ONE dword per coefficient (w0 | w1 << 24) -- no w1 byte plane
    int alloc(StreamScratch& ws, int level, int K, int L, size_t batch)
    {
        y = ws.take<int32_t>(batch * L * 256);
        w1 = ws.take<uint8_t>(batch * K * 256);
        w0 = ws.take<int32_t>(batch * K * 256);
        w1p = ws.take<uint8_t>(batch * K * (level == 2 ? 192 : 128));
        c = ws.take<int32_t>(batch * 256);
        return ws.rc;
    }
};
// A matrix per item travels through HBM as 24-bit packed coefficients (kernels.hpp A_P24) when the batch is large enough for
// the throughput form of ExpandA; one shared matrix, small batches and everything in the public API stay int32.
inline int matrix_format(size_t nkeys, int K, int L)
{
    return (nkeys > 1 && nkeys * (size_t)(K * L) > dil::EA_TWO_LANE_MAX && dil::rt::cfg.a24.load(std::memory_order_relaxed)) ? dil::A_P24 : dil::A_I32;
}
// Run an independent part of a composite call on the helper stream (if nobody else is using it): fork() returns the
// stream to launch that part on -- the helper, ordered after everything already on `main`, or `main` itself -- and
// join() makes `main` wait for it.
struct AuxFork {
    Device& dv;
    std::unique_lock<std::mutex> lk;
    hipStream_t main;
    bool on, forked = false;
    AuxFork(Device& d, hipStream_t m) : dv(d), lk(d.aux.mu, std::try_to_lock), main(m)
    {
        on = dil::rt::cfg.aux_overlap.load(std::memory_order_relaxed) && lk.owns_lock() && dv.aux.ensure();
    }
    // defer = true: an inert object (never locks, never forks) -- the caller already owns the helper stream through another AuxFork
    AuxFork(Device& d, hipStream_t m, bool defer) : dv(d), lk(d.aux.mu, std::defer_lock), main(m)
    {
        if (!defer) {
            (void)lk.try_lock();
            on = dil::rt::cfg.aux_overlap.load(std::memory_order_relaxed) && lk.owns_lock() && dv.aux.ensure();
        } else {
            on = false;
        }
    }
    // `sponges`: lanes of the lane-per-sponge work going to the helper.  Only latency-bound work (less than about one
    // wave per SIMD) gains from running beside the main stream; throughput-bound work just pays the fork/join.
    hipStream_t fork(size_t sponges)
    {
        if (!on || sponges >= (size_t)dv.num_cus * 256) return main;
        if (hipEventRecord(dv.aux.fork, main) != hipSuccess || hipStreamWaitEvent(dv.aux.s, dv.aux.fork, 0) != hipSuccess) {
            on = false;
            return main;
        }
        forked = true;
        return dv.aux.s;
    }
    int join()
    {
        if (!forked) return 0;
        forked = false;
        DIL_TRY(hipEventRecord(dv.aux.join, dv.aux.s));
        DIL_TRY(hipStreamWaitEvent(main, dv.aux.join, 0));
        return 0;
    }
    ~AuxFork() { (void)join(); }
};
int sign_attempt_impl(const dil::Tables& T, const AttemptScratch& t, uint8_t* ctilde, int32_t* z, uint8_t* h, int32_t* flags, const int32_t* A,
                      const uint8_t* mu, const uint8_t* rhoprime, const uint32_t* kappa, const int32_t* s1hat, const int32_t* s2hat,
                      const int32_t* t0hat, int level, size_t batch, int shared_key, hipStream_t s, dil::KeyMap km = dil::KeyMap(),
                      bool early_exit = false)
{
    // y: int32, or -- every round wide enough for the wave-per-item kernels -- the raw B-bit SHAKE256 stream, unpacked by phase 1 /
    // phase 2 as they load it (ExpandMask picks one or two lanes per sponge by the round's width either way)
    const int y_fmt = (t.packed_y && dil::fused_wpi_shape(batch, T)) ? dil::Y_PACKED : dil::Y_I32;
    if (y_fmt == dil::Y_PACKED) DIL_TRY(dil::launch_expand_mask_packed(reinterpret_cast<uint8_t*>(t.y), rhoprime, kappa, level, batch, s));
    else DIL_TRY(dil::launch_expand_mask(t.y, rhoprime, kappa, level, batch, s));
    // phase 1 writes w1 packed (the challenge hash's input) and, for phase 2, either as a byte plane of its own (the public form) or -- in
    // the signing loop's wave-per-item rounds -- in the top byte of the w0 dwords (pipeline_common.hpp emit_matvec_row: the W0W1 plane)
    uint8_t* w1_plane = (t.w0w1 && dil::fused_wpi_shape(batch, T)) ? nullptr : t.w1;
    DIL_TRY(dil::launch_matvec(level, dil::OUT_W1W0, nullptr, w1_plane, t.w0, A, t.y, batch, shared_key, T, s, km, t.w1p, t.a_fmt, y_fmt));
    if (t.fuse_challenge) {
        DIL_TRY(dil::launch_challenge_sample(ctilde, t.c, mu, t.w1p, level, batch, s));
    } else {
        DIL_TRY(dil::launch_challenge_hash(ctilde, nullptr, mu, t.w1p, level, nullptr, batch, s));
        DIL_TRY(dil::launch_sample_in_ball(t.c, ctilde, level, batch, s));
    }
    DIL_TRY(dil::launch_sign2(level, z, h, flags, t.c, t.y, t.w0, w1_plane, s1hat, s2hat, t0hat, batch, shared_key, T, s, km,
                              early_exit ? t.w0 : nullptr, y_fmt, t.small_key));
    return 0;
}

}  // namespace

int dil_sign_attempt_dev(uint8_t* ctilde, int32_t* z, uint8_t* h, int32_t* flags, const int32_t* A, const uint8_t* mu,
                         const uint8_t* rhoprime, const uint32_t* kappa, const int32_t* s1hat, const int32_t* s2hat,
                         const int32_t* t0hat, int level, size_t batch, int shared_key, void* stream)
{
    int rc, K, L;
    if ((rc = level_kl(level, &K, &L))) return rc;
    DIL_ENTER(dv, T);
    if (batch == 0) return 0;
    hipStream_t s = S(stream);
    StreamScratch ws(dv, s);
    AttemptScratch t;
    t.fuse_challenge = dil::rt::cfg.fuse_challenge.load(std::memory_order_relaxed) != 0;
    if ((rc = t.alloc(ws, level, K, L, batch))) return rc;
    return ws.close(sign_attempt_impl(T, t, ctilde, z, h, flags, A, mu, rhoprime, kappa, s1hat, s2hat, t0hat, level, batch, shared_key, s));
}

// ---- rows N2 / N4: codecs, keygen, wire-format verify -------------------------------------------------
namespace {
struct LevelPar { int K, L, eta, omega, zbits, eta_bits; int32_t gamma1; };
int level_par(int level, LevelPar* p)
{
    switch (level) {
    case 2: *p = {4, 4, 2, 80, 18, 3, 1 << 17}; return 0;
    case 3: *p = {6, 5, 4, 55, 20, 4, 1 << 19}; return 0;
    case 5: *p = {8, 7, 2, 75, 20, 3, 1 << 19}; return 0;
    default: return (int)hipErrorInvalidValue;
    }
}
struct CodecDesc { int bits, polys, xf; int32_t offset; };
int codec_desc(int kind, const LevelPar& p, CodecDesc* d)
{
    switch (kind) {
    case DIL_CODEC_T1: *d = {10, p.K, dil::XF_PLAIN, 0}; return 0;
    case DIL_CODEC_T0: *d = {13, p.K, dil::XF_OFFSET_MINUS, 1 << 12}; return 0;
    case DIL_CODEC_S1: *d = {p.eta_bits, p.L, dil::XF_OFFSET_MINUS, p.eta}; return 0;
    case DIL_CODEC_S2: *d = {p.eta_bits, p.K, dil::XF_OFFSET_MINUS, p.eta}; return 0;
    case DIL_CODEC_Z: *d = {p.zbits, p.L, dil::XF_OFFSET_MINUS, p.gamma1}; return 0;
    default: return (int)hipErrorInvalidValue;
    }
}
}  // namespace

size_t dil_pk_bytes(int level) { LevelPar p; return level_par(level, &p) ? 0 : 32 + (size_t)p.K * 320; }
size_t dil_sk_bytes(int level)
{
    LevelPar p;
    return level_par(level, &p) ? 0 : 96 + (size_t)(p.L + p.K) * 32 * p.eta_bits + (size_t)p.K * 416;
}
size_t dil_sig_bytes(int level) { LevelPar p; return level_par(level, &p) ? 0 : 32 + (size_t)p.L * 32 * p.zbits + p.omega + p.K; }

int dil_unpack_dev(int32_t* out, const uint8_t* in, size_t in_stride, size_t in_offset, int kind, int level, size_t batch, void* stream)
{
    LevelPar p;
    CodecDesc d;
    int rc;
    if ((rc = level_par(level, &p)) || (rc = codec_desc(kind, p, &d))) return rc;
    DIL_ENTER(dv, T);
    return (int)dil::launch_unpack(d.bits, out, in, in_stride, in_offset, d.polys, d.xf, d.offset, batch, T, S(stream));
}
int dil_pack_dev(uint8_t* out, size_t out_stride, size_t out_offset, const int32_t* in, int kind, int level, size_t batch, void* stream)
{
    LevelPar p;
    CodecDesc d;
    int rc;
    if ((rc = level_par(level, &p)) || (rc = codec_desc(kind, p, &d))) return rc;
    DIL_ENTER(dv, T);
    return (int)dil::launch_pack(d.bits, out, out_stride, out_offset, in, d.polys, d.xf, d.offset, batch, T, S(stream));
}
int dil_hint_unpack_dev(uint8_t* h, int32_t* bad, const uint8_t* in, size_t in_stride, size_t in_offset, int level, size_t batch, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p))) return rc;
    DIL_ENTER(dv, T);
    return (int)dil::launch_hint_unpack(h, bad, in, in_stride, in_offset, p.K, p.omega, batch, S(stream));
}
int dil_hint_pack_dev(uint8_t* out, size_t out_stride, size_t out_offset, const uint8_t* h, int level, size_t batch, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p))) return rc;
    DIL_ENTER(dv, T);
    return (int)dil::launch_hint_pack(out, out_stride, out_offset, h, p.K, p.omega, batch, S(stream));
}
int dil_expand_s_dev(int32_t* s1, int32_t* s2, const uint8_t* rhoprime, size_t stride, int level, size_t batch, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p))) return rc;
    DIL_ENTER(dv, T);
    return (int)dil::launch_expand_s(s1, s2, rhoprime, stride, p.eta, p.L, p.K, batch, S(stream));
}

int dil_keygen_dev(uint8_t* pk, uint8_t* sk, const uint8_t* seed, int level, size_t batch, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p))) return rc;
    DIL_ENTER(dv, T);
    if (batch == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(pk) | reinterpret_cast<uintptr_t>(seed)) & 7) return (int)hipErrorInvalidValue;   // hashed as 64-bit words
    hipStream_t s = S(stream);
    StreamScratch ws(dv, s);
    const size_t pkb = dil_pk_bytes(level), skb = dil_sk_bytes(level), sb = (size_t)32 * p.eta_bits;
    uint8_t* e = ws.take<uint8_t>(batch * 128);                    // rho(32) | rho'(64) | key(32)  (KG_*)
    int32_t* A = ws.take<int32_t>(batch * p.K * p.L * 256);
    int32_t* s1 = ws.take<int32_t>(batch * p.L * 256);
    int32_t* s2 = ws.take<int32_t>(batch * p.K * 256);
    // fused output stage (large batches): t1 / t0 leave the mat-vec kernel packed, straight into pk / sk
    const bool fused = dil::keygen_fused_available(batch, T) && !(reinterpret_cast<uintptr_t>(sk) & 3) &&
                       dil::rt::cfg.fuse_keygen.load(std::memory_order_relaxed);
    int32_t* w = fused ? nullptr : ws.take<int32_t>(batch * p.K * 256);
    int32_t* t1 = fused ? nullptr : ws.take<int32_t>(batch * p.K * 256);
    int32_t* t0 = fused ? nullptr : ws.take<int32_t>(batch * p.K * 256);
    uint8_t* tr = ws.take<uint8_t>(batch * 32);
    if (ws.rc) return ws.rc;
    ws.secret = true;            // rho', key, s1, s2, t0
    AuxFork ax(dv, s);
    DIL_TRY(dil::launch_shake256(reinterpret_cast<uint64_t*>(e), 128, reinterpret_cast<const uint64_t*>(seed), 32, batch, s));
    const int a_fmt = matrix_format(batch, p.K, p.L);
    if (batch * p.K * p.L <= dil::EA_TWO_LANE_MAX && dil::rt::cfg.aux_overlap.load(std::memory_order_relaxed)) {
        // few keys: ExpandA and ExpandS are latency-bound two-lane sponges -- side by side in one launch
        DIL_TRY(dil::launch_expand_a_s(A, e, 128, s1, s2, e + 32, 128, level, p.eta, batch, s));
    } else {
        // ExpandS (helper stream, when it is latency-bound) runs beside ExpandA: independent, both Keccak-bound
        DIL_TRY(dil::launch_expand_s(s1, s2, e + 32, 128, p.eta, p.L, p.K, batch, ax.fork(batch * p.K * p.L)));
        DIL_TRY(dil::launch_expand_a(A, e, 128, level, batch, s, a_fmt));
        if ((rc = ax.join())) return rc;
    }
    // pk = rho | t1
    DIL_TRY(dil::launch_copy_field(pk, pkb, 0, e, 128, 0, 32, batch, T, s));
    if (fused) {
        DIL_TRY(dil::launch_keygen_matvec(level, pk, pkb, sk, skb, 96 + (p.L + p.K) * sb, A, s1, s2, batch, T, s, a_fmt));
    } else {
        DIL_TRY(dil::launch_matvec(level, dil::OUT_W, w, nullptr, nullptr, A, s1, batch, 0, T, s, dil::KeyMap(), nullptr, a_fmt));
        DIL_TRY(dil::launch_power2round(t1, t0, w, s2, batch * p.K * 256, T, s));
        DIL_TRY(dil::launch_pack(10, pk, pkb, 32, t1, p.K, dil::XF_PLAIN, 0, batch, T, s));
    }
    if (fused || (!(reinterpret_cast<uintptr_t>(sk) & 3) && dil::rt::cfg.fuse_keygen.load(std::memory_order_relaxed))) {
        // the rest of sk in ONE launch: tr = SHAKE256(pk, 32) (a long two-lane sponge per key) beside the copies of rho / key
        // and the packing of s1 / s2 (codec_kernels.hip keygen_finish_kernel) -- no helper stream.  (Small batches, whose
        // mat-vec ran unfused, still pack t0 with the codec kernel: a different region of sk.)
        if (!fused) DIL_TRY(dil::launch_pack(13, sk, skb, 96 + (p.L + p.K) * sb, t0, p.K, dil::XF_OFFSET_MINUS, 1 << 12, batch, T, s));
        DIL_TRY(dil::launch_keygen_finish(sk, skb, pk, pkb, e, s1, s2, p.L, p.K, p.eta, p.eta_bits, batch, s));
        return ws.close(ax.join());
    }
    // tr = SHAKE256(pk, 32) (pk length is a multiple of 8 at every level): one long sponge per key, latency-bound --
    // on the helper stream, under the packing of the rest of sk
    {
        hipStream_t a = ax.fork(batch);
        DIL_TRY(dil::launch_shake256(reinterpret_cast<uint64_t*>(tr), 32, reinterpret_cast<const uint64_t*>(pk), (int)pkb, batch, a));
        DIL_TRY(dil::launch_copy_field(sk, skb, 64, tr, 32, 0, 32, batch, T, a));
    }
    // sk = rho | key | tr | s1 | s2 | t0
    DIL_TRY(dil::launch_copy_field(sk, skb, 0, e, 128, 0, 32, batch, T, s));
    DIL_TRY(dil::launch_copy_field(sk, skb, 32, e, 128, 96, 32, batch, T, s));
    DIL_TRY(dil::launch_pack(p.eta_bits, sk, skb, 96, s1, p.L, dil::XF_OFFSET_MINUS, p.eta, batch, T, s));
    DIL_TRY(dil::launch_pack(p.eta_bits, sk, skb, 96 + p.L * sb, s2, p.K, dil::XF_OFFSET_MINUS, p.eta, batch, T, s));
    DIL_TRY(dil::launch_pack(13, sk, skb, 96 + (p.L + p.K) * sb, t0, p.K, dil::XF_OFFSET_MINUS, 1 << 12, batch, T, s));
    return ws.close(ax.join());
}

// Everything of a wire-format verification between ExpandA and the challenge hash as ONE fused kernel
// (wire_kernels.hip): c = SampleInBall(c~) in compact form, then packed z / t1 / hints in, packed w1 + verdict bits out.
int dil_verify_wire_core_dev(uint8_t* w1_packed, int32_t* verdict, const int32_t* A, const uint8_t* pk, const uint8_t* sig, int level,
                             size_t batch, int shared_pk, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p))) return rc;
    DIL_ENTER(dv, T);
    if (batch == 0) return 0;
    if (!A) return (int)hipErrorInvalidValue;
    hipStream_t s = S(stream);
    StreamScratch ws(dv, s);
    const size_t pkb = dil_pk_bytes(level), sgb = dil_sig_bytes(level);
    // c sampled inside the kernel: one launch, no compact c through HBM -- levels 2 and 3 (76 - 80 against 84 - 86 us and 57 against 64 us per 8192);
    // at level 5 the kernel sits at its 168-register cap and the sampler's registers spill (134 against 130 us): the launch in front stays
    // (profiles/r06g_fuse_sib.txt)
    if (!shared_pk && level != 5 && (dil::rt::cfg.fuse_sib.load(std::memory_order_relaxed) & 1))
        return ws.close((int)dil::launch_verify_wire(level, w1_packed, verdict, A, pk, pkb, sig, sgb, nullptr, batch, 0, T, s));
    uint32_t* cbits = ws.take<uint32_t>(batch * 64);
    if (ws.rc) return ws.rc;
    DIL_TRY(dil::launch_sample_in_ball_bits(cbits, sig, sgb, level, batch, s));
    return ws.close((int)dil::launch_verify_wire(level, w1_packed, verdict, A, pk, pkb, sig, sgb, cbits, batch, shared_pk, T, s));
}

namespace {
// wire-format verification on a caller-provided scratch (dil_verify_sig_dev, dil_verify_msg_dev)
int verify_sig_core(Device& dv, const dil::Tables& T, StreamScratch& ws, int32_t* verdict, const uint8_t* pk, const uint8_t* sig,
                    const uint8_t* mu, int level, const LevelPar& p, size_t batch, int shared_pk, hipStream_t s,
                    const int32_t* A_ready = nullptr,      // A_ready: the caller's ExpandA(rho) of every key (dil_expand_a_dev), kept across calls
                    AuxFork* mu_pending = nullptr,         // mu is still being computed on the helper stream: joined before its first use
                    const int32_t* t1hat_ready = nullptr)  // with A_ready: the keys' NTT(t1 2^13) too (dil_expand_t1_dev), a key per item
{
    int rc;
    const size_t pkb = dil_pk_bytes(level), sgb = dil_sig_bytes(level), zb = (size_t)p.L * 32 * p.zbits;
    const size_t nk = shared_pk ? 1 : batch;
    const size_t w1b = (size_t)p.K * (level == 2 ? 192 : 128);
    const bool few_keys_path = dil::rt::cfg.fuse_wire.load(std::memory_order_relaxed) && !A_ready && nk * p.K * p.L <= dil::EA_TWO_LANE_MAX &&
                               dil::rt::cfg.aux_overlap.load(std::memory_order_relaxed);
    if (mu_pending && !few_keys_path && (rc = mu_pending->join())) return rc;      // only that path defers the join to mu's first use
    if (dil::rt::cfg.fuse_wire.load(std::memory_order_relaxed)) {
        // Fused path: ExpandA (helper stream when it is latency-bound) beside SampleInBall, then ONE kernel that reads
        // the packed z / t1 / hints and writes packed w1 (+ the ||z|| and hint-encoding verdict bits), then the challenge
        // hash compared with c~ in place.  No int32 z / t1 / h / c / w1 temporaries.
        // (Sampling A INSIDE the verifying kernel -- the reference's gen_a_ext.v feeding the MAC -- was built in round 2, measured slower
        //  than ExpandA -> HBM -> this kernel (297 vs 239 us per 8192, profiles/r02_gen_a.txt) and removed in round 4; so was a
        //  three-stream chunk pipeline of ExpandA / fused kernel / challenge hash (round 3, profiles/r03f_verify_chunks.txt).)
        int32_t* A = A_ready ? const_cast<int32_t*>(A_ready) : ws.take<int32_t>(nk * p.K * p.L * 256);
        uint32_t* cbits = ws.take<uint32_t>(batch * 64);
        uint8_t* w1p = ws.take<uint8_t>(batch * w1b);
        if (ws.rc) return ws.rc;
        const int fuse_sib = dil::rt::cfg.fuse_sib.load(std::memory_order_relaxed);
        if (A_ready) {               // the matrix is already there: SampleInBall (inside the fused kernel where that form exists), the fused kernel, the challenge hash
            const bool sib_inside = (fuse_sib & 1) && !shared_pk && !t1hat_ready && level != 5;
            if (!sib_inside) DIL_TRY(dil::launch_sample_in_ball_bits(cbits, sig, sgb, level, batch, s));
            DIL_TRY(dil::launch_verify_wire(level, w1p, verdict, A, pk, pkb, sig, sgb, sib_inside ? nullptr : cbits, batch, shared_pk, T, s, dil::A_I32, t1hat_ready));
            return (int)dil::launch_challenge_hash(nullptr, verdict, mu, w1p, level, sig, batch, s, sgb);
        }
        // Two independent Keccak jobs: ExpandA (nk * K * L sponges) and SampleInBall (batch sponges, one lane each).
        const size_t a_sp = nk * p.K * p.L;
        if (a_sp <= dil::EA_TWO_LANE_MAX && dil::rt::cfg.aux_overlap.load(std::memory_order_relaxed)) {
            // few keys: both jobs are latency-bound dependency chains -- ONE launch runs them side by side on different CUs
            // (wire_kernels.hip expand_a_sib_kernel; no helper stream, no fork / join events)
            DIL_TRY(dil::launch_expand_a_sib(A, pk, pkb, nk, cbits, sig, sgb, level, batch, s));
            DIL_TRY(dil::launch_verify_wire(level, w1p, verdict, A, pk, pkb, sig, sgb, cbits, batch, shared_pk, T, s));
            if (mu_pending && (rc = mu_pending->join())) return rc;
            return (int)dil::launch_challenge_hash(nullptr, verdict, mu, w1p, level, sig, batch, s, sgb);
        }
        // Many keys: the smaller job goes to the helper stream, under the larger one.  (Tried: SampleInBall riding in the
        // first workgroups of the throughput ExpandA's launch -- level 2 +8 %, level 3 -1.5 %, level 5 +2.5 % at 8192 keys,
        // -3 % at 65536 keys at every level: not kept.)
        // (the helper stream's owner is whoever holds dv.aux.mu: a caller that already forked on it -- dil_verify_msg_dev, whose mu
        //  chain was joined above -- hands its AuxFork down; constructing a second one on this thread would try_lock a mutex the
        //  thread owns, which is undefined behaviour and, on glibc, silently turned the overlap off)
        AuxFork own_ax(dv, s, /*defer=*/mu_pending != nullptr);
        AuxFork& ax = mu_pending ? *mu_pending : own_ax;
        const size_t a_sponges = nk * p.K * p.L;
        // (int32 matrix here: the fused verify kernel is not bound by the A stream -- with 24-bit packed A it runs 64.7 vs
        //  63.1 us and ExpandA's 48-byte pieces cost 8 us more than its 64-byte ones; profiles/r02_a24.txt.  The format
        //  parameter stays for A/B runs: option a24 = 2 forces the packed form here too.)
        const int a_fmt = (!shared_pk && dil::rt::cfg.a24.load(std::memory_order_relaxed) == 2) ? matrix_format(nk, p.K, p.L) : dil::A_I32;
        if ((fuse_sib & 2) && !shared_pk && a_fmt == dil::A_I32 && level != 5) {      // c inside the fused kernel: ExpandA alone in front, no helper stream
            DIL_TRY(dil::launch_expand_a(A, pk, pkb, level, nk, s, a_fmt));
            DIL_TRY(dil::launch_verify_wire(level, w1p, verdict, A, pk, pkb, sig, sgb, nullptr, batch, 0, T, s, a_fmt));
            return (int)dil::launch_challenge_hash(nullptr, verdict, mu, w1p, level, sig, batch, s, sgb);
        }
        hipStream_t sa = a_sponges <= batch ? ax.fork(a_sponges) : s;
        hipStream_t sc = a_sponges <= batch ? s : ax.fork(batch);
        DIL_TRY(dil::launch_expand_a(A, pk, pkb, level, nk, sa, a_fmt));
        DIL_TRY(dil::launch_sample_in_ball_bits(cbits, sig, sgb, level, batch, sc));
        if ((rc = ax.join())) return rc;
        DIL_TRY(dil::launch_verify_wire(level, w1p, verdict, A, pk, pkb, sig, sgb, cbits, batch, shared_pk, T, s, a_fmt));
        return (int)dil::launch_challenge_hash(nullptr, verdict, mu, w1p, level, sig, batch, s, sgb);
    }
    int32_t* A = ws.take<int32_t>(nk * p.K * p.L * 256);
    int32_t* t1 = ws.take<int32_t>(nk * p.K * 256);
    int32_t* z = ws.take<int32_t>(batch * p.L * 256);
    uint8_t* h = ws.take<uint8_t>(batch * p.K * 256);
    int32_t* bad = ws.take<int32_t>(batch);
    uint8_t* ct = ws.take<uint8_t>(batch * 32);
    int32_t* c = ws.take<int32_t>(batch * 256);
    uint8_t* w1 = ws.take<uint8_t>(batch * p.K * 256);
    uint8_t* w1p = ws.take<uint8_t>(batch * w1b);
    if (ws.rc) return ws.rc;
    AuxFork own_ax(dv, s, /*defer=*/mu_pending != nullptr);
    AuxFork& ax = mu_pending ? *mu_pending : own_ax;
    {   // public-key side (helper stream when it is latency-bound): A = ExpandA(rho), t1
        hipStream_t a = ax.fork(nk * p.K * p.L);
        DIL_TRY(dil::launch_expand_a(A, pk, pkb, level, nk, a));
        DIL_TRY(dil::launch_unpack(10, t1, pk, pkb, 32, p.K, dil::XF_PLAIN, 0, nk, T, a));
    }
    // signature side: c~, z, hints, ||z|| check, c = SampleInBall(c~)
    DIL_TRY(dil::launch_copy_field(ct, 32, 0, sig, sgb, 0, 32, batch, T, s));
    DIL_TRY(dil::launch_unpack(p.zbits, z, sig, sgb, 32, p.L, dil::XF_OFFSET_MINUS, p.gamma1, batch, T, s));
    DIL_TRY(dil::launch_hint_unpack(h, bad, sig, sgb, 32 + zb, p.K, p.omega, batch, s));
    DIL_TRY(dil::launch_z_norm(verdict, z, level, batch, s));
    DIL_TRY(dil::launch_sample_in_ball(c, ct, level, batch, s));
    if ((rc = ax.join())) return rc;
    DIL_TRY(dil::launch_verify(level, w1, A, z, c, t1, h, batch, shared_pk, T, s));
    DIL_TRY(dil::launch_pack_w1(w1p, w1, level, batch, T, s));
    DIL_TRY(dil::launch_challenge_hash(nullptr, verdict, mu, w1p, level, ct, batch, s));
    return (int)dil::launch_or_flag(verdict, bad, 4, batch, T, s);
}
}  // namespace

int dil_verify_sig_dev(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level, size_t batch,
                       int shared_pk, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p))) return rc;
    DIL_ENTER(dv, T);
    if (batch == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(pk) | reinterpret_cast<uintptr_t>(mu)) & 7) return (int)hipErrorInvalidValue;   // read as 64-bit words
    hipStream_t s = S(stream);
    StreamScratch ws(dv, s);
    return ws.close(verify_sig_core(dv, T, ws, verdict, pk, sig, mu, level, p, batch, shared_pk, s));
}


// Verification against keys whose matrix the caller has expanded once and keeps (dil_expand_a_dev on the keys' rho): the
// case of many signatures under few public keys arriving over many calls.  Skips ExpandA -- 184 of the 283 us a level-3
// batch of 8192 costs with a key per signature, 47 of 157 us with one key.  Always the fused wire-format kernel.
int dil_verify_sig_expanded_dev(int32_t* verdict, const int32_t* A, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level,
                                size_t batch, int shared_pk, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p))) return rc;
    DIL_ENTER(dv, T);
    if (batch == 0) return 0;
    if (!A || (reinterpret_cast<uintptr_t>(mu) & 7) || (reinterpret_cast<uintptr_t>(A) & 15)) return (int)hipErrorInvalidValue;
    hipStream_t s = S(stream);
    StreamScratch ws(dv, s);
    const int fuse = dil::rt::cfg.fuse_wire.load(std::memory_order_relaxed);
    if (!fuse) return (int)hipErrorNotSupported;             // this entry point exists only in the fused form
    return ws.close(verify_sig_core(dv, T, ws, verdict, pk, sig, mu, level, p, batch, shared_pk, s, A));
}

// The same with t1^ = NTT(t1 2^13) of every key kept beside its matrix (dil_expand_t1_dev): the fused kernel runs L + 1 forward and K
// inverse transforms per verification instead of L + 1 + K and K (VY_NTT_T1, combined_top.v:1259-1313, once per key).  t1hat is used
// with a key per signature; with one key for the batch the shared-key kernel keeps t1^ in LDS anyway and t1hat is ignored.
int dil_verify_sig_expanded2_dev(int32_t* verdict, const int32_t* A, const int32_t* t1hat, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu,
                                 int level, size_t batch, int shared_pk, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p))) return rc;
    DIL_ENTER(dv, T);
    if (batch == 0) return 0;
    if (!A || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(mu) & 7)) return (int)hipErrorInvalidValue;
    if (!shared_pk && (!t1hat || (reinterpret_cast<uintptr_t>(t1hat) & 15))) return (int)hipErrorInvalidValue;     // (one key for the batch: t1hat is not read)
    hipStream_t s = S(stream);
    StreamScratch ws(dv, s);
    if (!dil::rt::cfg.fuse_wire.load(std::memory_order_relaxed)) return (int)hipErrorNotSupported;
    return ws.close(verify_sig_core(dv, T, ws, verdict, pk, sig, mu, level, p, batch, shared_pk, s, A, nullptr, shared_pk ? nullptr : t1hat));
}
int dil_expand_t1_dev(int32_t* t1hat, const uint8_t* pk, int level, size_t nkeys, void* stream)
{
    LevelPar p;
    if (int rc = level_par(level, &p)) return rc;
    DIL_ENTER(dv, T);
    if (nkeys == 0) return 0;
    if (!t1hat || !pk || (reinterpret_cast<uintptr_t>(t1hat) & 15)) return (int)hipErrorInvalidValue;
    return (int)dil::launch_expand_t1(t1hat, pk, dil_pk_bytes(level), level, nkeys, T, S(stream));
}

// ---- row N3: the whole signing rejection loop on the device ---------------------------------------
// combined_top.v's sign FSMs (:1694-2229) retry one signature until it passes.  A batch engine is
// better used WIDE than deep: each round runs S speculative attempts (kappa = a0*L, (a0+1)*L, ...)
// for every still-pending signature, S chosen so that a round keeps about `cap` entries in flight;
// the first accepted attempt of an item wins, which is exactly the signature the sequential loop
// produces.  The pending set shrinks geometrically while S grows, so the loop needs ~5 rounds
// instead of the ~35 the unluckiest signature of a large batch takes.
namespace {
constexpr int SIGN_S_MAX = 64;      // attempts per item and round; also the wave width: phase 2 reads an item's earlier attempts one per lane (launch_sign2 refuses more)
// Entries kept in flight per round: the hash kernels are latency-bound below ~1 wave per SIMD, so small batches speculate for free
// (a round never uses more than batch * SIGN_S_MAX entries: a single signature gets 64 entries, not 16384).
// Default width: about one expected signature's worth of attempts per item in the first round (mean attempts 4.3 / 5.1 /
// 3.9 at levels 2 / 3 / 5), between 16384 and 32768 entries -- below that the round's kernels sit on their latency
// floors anyway, above it the speculation wastes more than a saved round is worth (scripts/bench_sign_cap.py,
// profiles/r02_sign_round.txt: level 3, 8192 messages 1.69 -> 1.57 ms with 24576 entries; 65536 messages: width = batch)
size_t sign_round_cap(int level, size_t batch)
{
    const int opt_cap = dil::rt::cfg.sign_cap.load(std::memory_order_relaxed);
    const size_t s0 = level == 2 ? 4 : level == 3 ? 3 : 2;
    const size_t dflt_cap = std::min<size_t>(std::max<size_t>(batch * s0, 16384), 32768);
    return std::min<size_t>(std::max<size_t>(batch, opt_cap > 0 ? (size_t)opt_cap : dflt_cap), batch * (size_t)SIGN_S_MAX);
}
// Attempts per pending item this round: as many as fit in `cap` entries, but only while the work expected to be
// wasted on attempts after an item's first success, n * (1 - (1-p)^(S-1)) entries with p ~ 0.2, stays below the
// work a saved round's fixed latency is worth (option sign_waste entries; matters for batches >> 16384).
int sign_round_width(size_t pending, size_t cap, int attempts_left)
{
    const int sign_waste = dil::rt::cfg.sign_waste.load(std::memory_order_relaxed);
    const int s_lim = (int)std::min<size_t>(std::min<size_t>(cap / pending, (size_t)SIGN_S_MAX), (size_t)attempts_left);
    int S_ = 1;
    double keep = 1.0;                       // (1-p)^(S-1)
    while (S_ < s_lim) {
        keep *= 0.8;
        if ((double)pending * (1.0 - keep) > (double)sign_waste) break;
        S_++;
    }
    return S_;
}

// Poll the mapped words for the sequence number the collect kernel posts behind its counts.  Every ~16 k polls the stream is asked
// whether it is still busy: a stream that has drained (or failed) without the number means the post never comes -- then the counts are
// fetched by an ordinary blocking copy (or the stream's error is returned), so a lost post costs time, never a hang.
int await_round_count(int32_t* host_words, const int32_t* counts, uint32_t seq, hipStream_t s)
{
    const uint32_t* posted = reinterpret_cast<const uint32_t*>(host_words + 2);
    for (unsigned polls = 1;; polls++) {
        if (__atomic_load_n(posted, __ATOMIC_ACQUIRE) == seq) return 0;
        if ((polls & 0x3fff) == 0) {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) {
                if (__atomic_load_n(posted, __ATOMIC_ACQUIRE) == seq) return 0;
                return (int)hipMemcpy(host_words, counts, 8, hipMemcpyDeviceToHost);
            }
            if (q != hipErrorNotReady) return (int)q;
        }
        __builtin_ia32_pause();
    }
}

int sign_core(Device& dv, const dil::Tables& T, StreamScratch& ws, uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* mu,
              int level, const LevelPar& p, size_t batch, int shared_sk, int max_attempts, hipStream_t s)
{
    int rc;
    if (batch == 1) shared_sk = 1;
    const size_t skb = dil_sk_bytes(level), sgb = dil_sig_bytes(level), zb = (size_t)p.L * 32 * p.zbits;
    const size_t nk = shared_sk ? 1 : batch, sk_stride = shared_sk ? 0 : skb;
    const bool sign_early = dil::rt::cfg.sign_early.load(std::memory_order_relaxed) != 0;
    const int sign_skip = dil::rt::cfg.sign_skip.load(std::memory_order_relaxed);       // bit 0: drop superseded attempts, bit 1: work queue
    const size_t cap = sign_round_cap(level, batch);
    ws.secret = true;            // s1^ s2^ t0^, key, rho', y, rejected z: wiped on close when option `zeroize` is set
    // per key
    int32_t* A = ws.take<int32_t>(nk * p.K * p.L * 256);
    int32_t* s1h = ws.take<int32_t>(nk * p.L * 256);
    int32_t* s2h = ws.take<int32_t>(nk * p.K * 256);
    int32_t* t0h = ws.take<int32_t>(nk * p.K * 256);
    // per item
    uint8_t* rp = ws.take<uint8_t>(batch * 64);          // rho'
    int32_t* idx0 = ws.take<int32_t>(batch);             // pending lists (ping-pong)
    int32_t* idx1 = ws.take<int32_t>(batch);
    int32_t* wine = ws.take<int32_t>(batch);             // winners of a round: entry, item
    int32_t* wini = ws.take<int32_t>(batch);
    int32_t* own_attempts = attempts ? nullptr : ws.take<int32_t>(batch);
    int32_t* counts = ws.take<int32_t>(4);               // [0] pending, [1] winners, [2] workgroups of the collect kernel that are through
    uint32_t* tickets = ws.take<uint32_t>(dil::TICKET_WORDS);      // phase 2's work queues (KeyMap::ticket)
    // per entry
    uint32_t* kap = ws.take<uint32_t>(cap);
    uint8_t* ct = ws.take<uint8_t>(cap * 32);
    int32_t* z = ws.take<int32_t>(cap * p.L * 256);
    uint8_t* h = ws.take<uint8_t>(cap * p.K * 256);
    int32_t* fl = ws.take<int32_t>(cap);
    uint8_t* mu_c = ws.take<uint8_t>(cap * 64);
    uint8_t* rp_c = ws.take<uint8_t>(cap * 64);
    AttemptScratch att;
    if ((rc = att.alloc(ws, level, p.K, p.L, cap))) return rc;
    if (!attempts) attempts = own_attempts;
    int32_t* host_counts = ws.pinned_words();            // pageable memory would make the read-back a blocking staged copy
    if (ws.rc) return ws.rc;

    {
        // Set-up in one launch (wire_kernels.hip sign_setup_kernel): s1^ s2^ t0^ = NTT(unpack(sk)) read from the packed key,
        // rho' = SHAKE256(key || mu, 64) with the key read in place (deterministic signing, as the reference's KATs), the attempt
        // counters cleared -- and, when the keys are few, A = ExpandA(rho) by the same launch's first workgroups (latency-bound,
        // two lanes per sponge).  Many keys: the throughput ExpandA first, on the same stream.
        att.a_fmt = matrix_format(nk, p.K, p.L);
        att.packed_y = dil::rt::cfg.packed_y.load(std::memory_order_relaxed) != 0;
        att.fuse_challenge = dil::rt::cfg.fuse_challenge.load(std::memory_order_relaxed) != 0;
        att.w0w1 = dil::rt::cfg.w0w1_plane.load(std::memory_order_relaxed) != 0;
        att.small_key = true;            // s1^ s2^ t0^ come from unpack(sk) and c from SampleInBall: the small-product kernels are exact (pipelines.hip)
        const bool few = nk * p.K * p.L <= dil::EA_TWO_LANE_MAX;            // (then the matrix format is int32)
        if (!few) DIL_TRY(dil::launch_expand_a(A, sk, skb, level, nk, s, att.a_fmt));
        DIL_TRY(dil::launch_sign_setup(level, A, few, s1h, s2h, t0h, sk, nk, rp, attempts, mu, sk_stride, batch, T, s));
    }

    // How a round's pending count reaches the host (option sign_wake): 1 = the collect kernel's last workgroup posts it into mapped host words
    // and the host polls them -- no copy, no event, no wake-up, so the next round's launches are queued while the winners are still being
    // packed; 0 = an 8-byte copy behind the collect kernel + an event (rounds 2 - 5).
    const int sign_wake = dil::rt::cfg.sign_wake.load(std::memory_order_relaxed);
    const bool wake_flag = sign_wake != 0;
    const bool lose_post = sign_wake == 2;      // tests only: the host waits for a number the kernel never posts -- the stream drains, the counts come by the blocking copy
    struct EventGuard {
        hipEvent_t ev = nullptr;
        ~EventGuard() { if (ev) (void)hipEventDestroy(ev); }
    } counted;
    if (!wake_flag) DIL_TRY(hipEventCreateWithFlags(&counted.ev, hipEventDisableTiming));
    int32_t *idx_cur = nullptr, *idx_next = idx0;
    size_t n = batch;
    int a0 = 0;                                          // attempts every pending item has already failed
    while (n > 0 && a0 < max_attempts) {
        const int S_ = sign_round_width(n, cap, max_attempts - a0);
        const size_t E = n * (size_t)S_;
        const bool direct = !idx_cur && S_ == 1;         // first round of a full batch: the caller's arrays as they are
        const uint8_t *mur = direct ? mu : mu_c, *rpr = direct ? rp : rp_c;
        dil::KeyMap keys;                                // per-item keys are read in place through the pending list
        keys.idx = idx_cur;
        keys.S = (uint32_t)S_;
        // attempts behind an item's first accepted one are never used: phase 2 may drop them as it goes (its early-exit form only)
        keys.spec_n = (S_ > 1 && sign_early && (sign_skip & 1)) ? (uint32_t)n : 0;
        keys.ticket = (sign_early && (sign_skip & 2)) ? tickets : nullptr;
        // one launch: gathers of mu / rho' for the entries, kappa = (a0 + e % S) L, counters cleared
        if (((reinterpret_cast<uintptr_t>(mu)) & 15) == 0) {
            DIL_TRY(dil::launch_sign_round_setup(mu_c, rp_c, kap, fl, counts, tickets, mu, rp, idx_cur, (uint32_t)a0, (uint32_t)p.L, (uint32_t)S_, E,
                                                 !direct, s));
        } else {                                         // caller's mu only 8-byte aligned: the generic kernels
            if (!direct) {
                DIL_TRY(dil::launch_gather_rows(mu_c, mu, idx_cur, 64, (uint32_t)S_, E, T, s));
                DIL_TRY(dil::launch_gather_rows(rp_c, rp, idx_cur, 64, (uint32_t)S_, E, T, s));
            }
            DIL_TRY(dil::launch_sign_kappa(kap, fl, (uint32_t)a0, (uint32_t)p.L, (uint32_t)S_, E, s));
            DIL_TRY(hipMemsetAsync(counts, 0, 16, s));
            DIL_TRY(hipMemsetAsync(tickets, 0, dil::TICKET_WORDS * 4, s));
        }
        // (Two stream-level overlaps of a round's latency-bound hash kernels with its polynomial kernels were built in rounds 2 / 3 and
        //  measured slower -- 1.42 -> 1.77 ms per 8192 level-3 signatures, profiles/r03j_sign_overlap.txt -- and are gone.)
        if ((rc = sign_attempt_impl(T, att, ct, z, h, fl, A, mur, rpr, kap, s1h, s2h, t0h, level, E, shared_sk, s, keys, sign_early))) return rc;
        // winners (first accepted attempt per item) -> packed straight into their signature slots; c~ rides in the collect kernel
        uint32_t seq = 0;
        if (wake_flag) {
            uint32_t* host_seq = reinterpret_cast<uint32_t*>(host_counts + 3);          // the host's own counter (unsigned: wraps, never overflows)
            seq = ++*host_seq;
            if (seq == 0) seq = ++*host_seq;                         // (0 is what a fresh allocation shows)
        }
        DIL_TRY(dil::launch_sign_collect_ct(attempts, idx_next, wine, wini, counts, fl, idx_cur, a0, S_, n, sig, sgb, ct, s,
                                            wake_flag ? host_counts : nullptr, seq));
        // the pending count goes home NOW; the winners' packing is queued behind it, so the host sizes the next round and has its
        // launches in the queue while the packing kernels still run
        if (!wake_flag) {
            DIL_TRY(hipMemcpyAsync(host_counts, counts, 8, hipMemcpyDeviceToHost, s));
            DIL_TRY(hipEventRecord(counted.ev, s));
        }
        dil::RowMap win;
        win.src_row = wine;
        win.dst_row = wini;
        win.count = counts + 1;
        DIL_TRY(dil::launch_pack(p.zbits, sig, sgb, 32, z, p.L, dil::XF_OFFSET_MINUS, p.gamma1, n, T, s, win));
        DIL_TRY(dil::launch_hint_pack(sig, sgb, 32 + zb, h, p.K, p.omega, n, s, win));
        if (wake_flag) {
            if ((rc = await_round_count(host_counts, counts, lose_post ? seq ^ 0x80000000u : seq, s))) return rc;
        } else {
            DIL_TRY(hipEventSynchronize(counted.ev));
        }
        n = (size_t)host_counts[0];
        a0 += S_;
        idx_cur = idx_next;
        idx_next = idx_cur == idx0 ? idx1 : idx0;
    }
    return n == 0 ? 0 : DIL_ERR_UNFINISHED;
}
}  // namespace

int dil_sign_round_plan(int level, size_t batch, size_t pending, int attempts_done, int max_attempts, int* attempts_per_item, size_t* entries)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p))) return rc;
    if (batch == 0 || pending == 0 || pending > batch || attempts_done < 0 || max_attempts <= attempts_done) return (int)hipErrorInvalidValue;
    const int S_ = sign_round_width(pending, sign_round_cap(level, batch), max_attempts - attempts_done);
    if (attempts_per_item) *attempts_per_item = S_;
    if (entries) *entries = pending * (size_t)S_;
    return 0;
}

int dil_sign_dev(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* mu, int level, size_t batch, int shared_sk,
                 int max_attempts, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p))) return rc;
    DIL_ENTER(dv, T);
    if (batch == 0) return 0;
    if (batch > 0x3fffffffull || max_attempts <= 0) return (int)hipErrorInvalidValue;
    if ((reinterpret_cast<uintptr_t>(sk) | reinterpret_cast<uintptr_t>(mu)) & 7) return (int)hipErrorInvalidValue;   // hashed as 64-bit words
    hipStream_t s = S(stream);
    StreamScratch ws(dv, s);
    return ws.close(sign_core(dv, T, ws, sig, attempts, sk, mu, level, p, batch, shared_sk, max_attempts, s));
}

// ---- message hashing on the device: mu = SHAKE256(tr || M, 64) for ragged M, then the operations on (key, M[, sig]) ----
// What the reference's top level absorbs itself (rtl_src/expandmask_ext.v:131-185; bus order mlen, tr, m in
// rtl_tb/tb_sign_top.v:57-69 and tb_verify_top.v:58-68).
namespace {
int check_msgs(const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets, const uint32_t* lengths)
{
    if ((!msgs && msgs_bytes) || !offsets || !lengths) return (int)hipErrorInvalidValue;      // msgs == NULL: an empty blob (every message empty)
    if ((reinterpret_cast<uintptr_t>(offsets) & 7) || (reinterpret_cast<uintptr_t>(lengths) & 3)) return (int)hipErrorInvalidValue;
    return 0;
}
}  // namespace

int dil_mu_dev(uint8_t* mu, const uint8_t* tr, size_t tr_stride, const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets,
               const uint32_t* lengths, int32_t* bad, size_t batch, void* stream)
{
    int rc;
    if ((rc = check_msgs(msgs, msgs_bytes, offsets, lengths))) return rc;
    if ((reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(tr) | tr_stride) & 7) return (int)hipErrorInvalidValue;
    DIL_ENTER(dv, T);
    return (int)dil::launch_mu(mu, tr, tr_stride, msgs, msgs_bytes, offsets, lengths, bad, batch, S(stream));
}

int dil_sign_msg_dev(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets,
                     const uint32_t* lengths, int level, size_t batch, int shared_sk, int max_attempts, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p)) || (rc = check_msgs(msgs, msgs_bytes, offsets, lengths))) return rc;
    DIL_ENTER(dv, T);
    if (batch == 0) return 0;
    if (batch > 0x3fffffffull || max_attempts <= 0) return (int)hipErrorInvalidValue;
    if (reinterpret_cast<uintptr_t>(sk) & 7) return (int)hipErrorInvalidValue;
    // attempts[i] = -1 is the ONLY sign that item i's message reference left the blob (its signature bytes are zeroed): the call is
    // asynchronous and cannot turn device-side findings into a return code, so the array is mandatory here (round-3 advisor finding)
    if (!attempts) return (int)hipErrorInvalidValue;
    hipStream_t s = S(stream);
    StreamScratch ws(dv, s);
    uint8_t* mu = ws.take<uint8_t>(batch * 64);
    int32_t* bad = ws.take<int32_t>(batch);
    if (ws.rc) return ws.rc;
    // tr sits at byte 64 of the secret key (rho | key | tr | ...)
    DIL_TRY(dil::launch_mu(mu, sk + 64, (shared_sk || batch == 1) ? 0 : dil_sk_bytes(level), msgs, msgs_bytes, offsets, lengths, bad, batch, s));
    rc = sign_core(dv, T, ws, sig, attempts, sk, mu, level, p, batch, shared_sk, max_attempts, s);
    if (rc == 0 || rc == DIL_ERR_UNFINISHED) {        // items whose message reference left the blob: no signature, attempts = -1
        const hipError_t e = dil::launch_sign_void_bad(sig, dil_sig_bytes(level), attempts, bad, batch, s);
        if (e != hipSuccess) rc = (int)e;
    }
    return ws.close(rc);
}

int dil_verify_msg_dev(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* msgs, size_t msgs_bytes, const uint64_t* offsets,
                       const uint32_t* lengths, int level, size_t batch, int shared_pk, void* stream)
{
    LevelPar p;
    int rc;
    if ((rc = level_par(level, &p)) || (rc = check_msgs(msgs, msgs_bytes, offsets, lengths))) return rc;
    DIL_ENTER(dv, T);
    if (batch == 0) return 0;
    if (reinterpret_cast<uintptr_t>(pk) & 7) return (int)hipErrorInvalidValue;
    hipStream_t s = S(stream);
    StreamScratch ws(dv, s);
    const size_t nk = shared_pk ? 1 : batch, pkb = dil_pk_bytes(level);
    uint8_t* tr = ws.take<uint8_t>(nk * 32);
    uint8_t* mu = ws.take<uint8_t>(batch * 64);
    int32_t* bad = ws.take<int32_t>(batch);
    if (ws.rc) return ws.rc;
    // tr = SHAKE256(pk, 32) (pk length is a multiple of 8 at every level), then mu = SHAKE256(tr || M, 64): with few keys a
    // latency-bound chain (15 + permutations in a row) that nothing needs before the challenge hash at the very end -- it runs
    // on the helper stream beside ExpandA / SampleInBall / the fused kernel and is joined there
    AuxFork ax(dv, s);
    hipStream_t h = nk * p.K * p.L <= dil::EA_TWO_LANE_MAX ? ax.fork(nk) : s;
    DIL_TRY(dil::launch_shake256(reinterpret_cast<uint64_t*>(tr), 32, reinterpret_cast<const uint64_t*>(pk), (int)pkb, nk, h));
    DIL_TRY(dil::launch_mu(mu, tr, shared_pk ? 0 : 32, msgs, msgs_bytes, offsets, lengths, bad, batch, h));
    if ((rc = verify_sig_core(dv, T, ws, verdict, pk, sig, mu, level, p, batch, shared_pk, s, nullptr, &ax))) return rc;
    // (verify_sig_core joined the helper stream before the challenge hash, its last launch on `s`: bad[] is complete here)
    return ws.close((int)dil::launch_or_flag(verdict, bad, 8, batch, T, s));
}

// ---- host-buffer forms of the whole operations (H2D -> device call -> D2H on the null stream) --------
namespace {
// staging buffers of the host-buffer forms: their own arena (the device call they wrap owns the null stream's)
const hipStream_t HOST_STAGE_KEY = reinterpret_cast<hipStream_t>(static_cast<uintptr_t>(1));
}  // namespace

int dil_keygen_host(uint8_t* pk, uint8_t* sk, const uint8_t* seed, int level, size_t batch)
{
    const size_t pkb = dil_pk_bytes(level), skb = dil_sk_bytes(level);
    if (!pkb) return (int)hipErrorInvalidValue;
    if (batch == 0) return 0;
    int rc;
    DIL_ENTER(dv, T);
    StreamScratch ws(dv, HOST_STAGE_KEY, nullptr);
    uint8_t* dpk = ws.take<uint8_t>(batch * pkb);
    uint8_t* dsk = ws.take<uint8_t>(batch * skb);
    uint8_t* dseed = ws.take<uint8_t>(batch * 32);
    if (ws.rc) return ws.rc;
    ws.secret = true;            // seed and the staged secret keys
    if ((rc = dil::rt::host_upload(dv, dseed, seed, batch * 32))) return rc;
    rc = dil_keygen_dev(dpk, dsk, dseed, level, batch, nullptr);
    if (rc) return rc;
    if ((rc = dil::rt::host_download(dv, pk, dpk, batch * pkb))) return rc;
    if ((rc = dil::rt::host_download(dv, sk, dsk, batch * skb))) return rc;
    return ws.close(0);
}

int dil_sign_host(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* mu, int level, size_t batch, int shared_sk,
                  int max_attempts)
{
    const size_t skb = dil_sk_bytes(level), sgb = dil_sig_bytes(level);
    if (!skb) return (int)hipErrorInvalidValue;
    if (batch == 0) return 0;
    DIL_ENTER(dv, T);
    const size_t nk = shared_sk ? 1 : batch;
    StreamScratch ws(dv, HOST_STAGE_KEY, nullptr);
    uint8_t* dsig = ws.take<uint8_t>(batch * sgb);
    int32_t* datt = ws.take<int32_t>(batch);
    uint8_t* dsk = ws.take<uint8_t>(nk * skb);
    uint8_t* dmu = ws.take<uint8_t>(batch * 64);
    if (ws.rc) return ws.rc;
    ws.secret = true;            // the staged secret key
    int rc;
    if ((rc = dil::rt::host_upload(dv, dsk, sk, nk * skb))) return rc;
    if ((rc = dil::rt::host_upload(dv, dmu, mu, batch * 64))) return rc;
    const int src = dil_sign_dev(dsig, datt, dsk, dmu, level, batch, shared_sk, max_attempts, nullptr);
    if (src && src != DIL_ERR_UNFINISHED) return src;
    DIL_TRY(hipStreamSynchronize(nullptr));
    if ((rc = dil::rt::host_download(dv, sig, dsig, batch * sgb))) return rc;
    if (attempts && (rc = dil::rt::host_download(dv, attempts, datt, batch * 4))) return rc;
    return ws.close(src);
}

int dil_verify_sig_host(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level, size_t batch,
                        int shared_pk)
{
    const size_t pkb = dil_pk_bytes(level), sgb = dil_sig_bytes(level);
    if (!pkb) return (int)hipErrorInvalidValue;
    if (batch == 0) return 0;
    int rc;
    DIL_ENTER(dv, T);
    const size_t nk = shared_pk ? 1 : batch;
    StreamScratch ws(dv, HOST_STAGE_KEY, nullptr);
    int32_t* dverd = ws.take<int32_t>(batch);
    uint8_t* dpk = ws.take<uint8_t>(nk * pkb);
    uint8_t* dsig = ws.take<uint8_t>(batch * sgb);
    uint8_t* dmu = ws.take<uint8_t>(batch * 64);
    if (ws.rc) return ws.rc;
    if ((rc = dil::rt::host_upload(dv, dpk, pk, nk * pkb))) return rc;
    if ((rc = dil::rt::host_upload(dv, dsig, sig, batch * sgb))) return rc;
    if ((rc = dil::rt::host_upload(dv, dmu, mu, batch * 64))) return rc;
    rc = dil_verify_sig_dev(dverd, dpk, dsig, dmu, level, batch, shared_pk, nullptr);
    if (rc) return rc;
    if ((rc = dil::rt::host_download(dv, verdict, dverd, batch * 4))) return rc;
    return 0;
}
