// capi.hip -- the extern "C" boundary of libdil256.so (declared in include/dil256.h) and the
// host runtime behind it: twiddle-table construction, device selection, scratch management
// for the host-pointer entry points.  Host language is C++ because the reference's
// dilithium-256/ is C++; nothing here is a CPU fallback -- every arithmetic entry
// point launches a HIP kernel and returns the hipError_t if that is not possible.
#include "capi_internal.hpp"
#include "launch_util.hpp"
#include "copy_pool.hpp"

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <stdlib.h>
#include <string.h>
#include <time.h>

namespace {
constexpr int64_t Q = DIL_Q;
void build_tables(uint32_t* fwd, uint32_t* inv, uint32_t* inv_pipe);
void mailbox_destroy(dil::rt::Device& d);
}  // namespace

namespace dil {
namespace rt {
Config cfg;
namespace {
Device g_dev[MAX_DEVICES];
std::once_flag g_env_once;

struct OptName { const char* name; const char* env; std::atomic<int>* slot; };
const OptName* option_table(int* n)
{
    static const OptName tab[] = {
        {"fused_mode", "DIL_FUSED_MODE", &cfg.fused_mode},
        {"ntt_blocks_per_cu", "DIL_NTT_BPC", &cfg.ntt_blocks_per_cu},
        {"wpi_blocks_per_cu", "DIL_WPI_BPC", &cfg.wpi_blocks_per_cu},
        {"fused_wgs_per_cu", "DIL_FUSED_WGPC", &cfg.fused_wgs_per_cu},
        {"sign_early", "DIL_SIGN_EARLY", &cfg.sign_early},
        {"sign_waste", "DIL_SIGN_WASTE", &cfg.sign_waste},
        {"sign_skip", "DIL_SIGN_SKIP", &cfg.sign_skip},
        {"sign_cap", "DIL_SIGN_CAP", &cfg.sign_cap},
        {"sign_wake", "DIL_SIGN_WAKE", &cfg.sign_wake},
        {"aux_overlap", "DIL_AUX_OVERLAP", &cfg.aux_overlap},
        {"zeroize", "DIL_ZEROIZE", &cfg.zeroize},
        {"fuse_wire", "DIL_FUSE_WIRE", &cfg.fuse_wire},
        {"packed_y", "DIL_PACKED_Y", &cfg.packed_y},
        {"w0w1_plane", "DIL_W0W1_PLANE", &cfg.w0w1_plane},
        {"multi_group_at_1", "DIL_MULTI_GROUP_AT_1", &cfg.multi_group_at_1},
        {"host_chunk", "DIL_HOST_CHUNK", &cfg.host_chunk},
        {"host_chunk_pinned", "DIL_HOST_CHUNK_PINNED", &cfg.host_chunk},      // (rounds 4-5 kept a second chunk size for page-locked buffers: one now, both names)
        {"host_streams", "DIL_HOST_STREAMS", &cfg.host_streams},
        {"host_copy_threads", "DIL_HOST_COPY_THREADS", &cfg.host_copy_threads},
        {"host_duplex", "DIL_HOST_DUPLEX", &cfg.host_duplex},
        {"host_mailbox", "DIL_HOST_MAILBOX", &cfg.host_mailbox},
        {"mailbox_idle_us", "DIL_MAILBOX_IDLE_US", &cfg.mailbox_idle_us},
        {"mailbox_resident_us", "DIL_MAILBOX_RESIDENT_US", &cfg.mailbox_resident_us},
        {"fuse_challenge", "DIL_FUSE_CHALLENGE", &cfg.fuse_challenge},
        {"a24", "DIL_A24", &cfg.a24},
        {"fuse_keygen", "DIL_FUSE_KEYGEN", &cfg.fuse_keygen},
        {"fuse_sib", "DIL_FUSE_SIB", &cfg.fuse_sib},
        {"two_lane_max_sponges", "DIL_TWO_LANE_MAX", &dil::two_lane_max_sponges},
        {"coop_max", "DIL_COOP_MAX", &dil::coop_max_sponges},
    };
    *n = (int)(sizeof(tab) / sizeof(tab[0]));
    return tab;
}
void read_env()
{
    int n;
    const OptName* tab = option_table(&n);
    for (int i = 0; i < n; i++)
        if (const char* e = getenv(tab[i].env)) tab[i].slot->store(atoi(e));
}

// bring one device up: twiddle tables, CU count, private spill pool.  Caller holds d.mu and has `id` current.
int init_device(Device& d, int id)
{
    hipDeviceProp_t prop;
    DIL_TRY(hipGetDeviceProperties(&prop, id));
    static uint32_t h_tab[3 * 2048];
    static std::once_flag tab_once;
    std::call_once(tab_once, [] { build_tables(h_tab, h_tab + 2048, h_tab + 4096); });
    DIL_TRY(hipMalloc(reinterpret_cast<void**>(&d.d_tables), sizeof(h_tab)));
    DIL_TRY(hipMemcpy(d.d_tables, h_tab, sizeof(h_tab), hipMemcpyHostToDevice));
    d.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    d.id = id;
    {   // Spills of the composite calls come from a PRIVATE stream-ordered pool that keeps what it has grown to
        // (the application's default pool and its release threshold are not touched).
        hipMemPoolProps pp = {};
        pp.allocType = hipMemAllocationTypePinned;
        pp.location.type = hipMemLocationTypeDevice;
        pp.location.id = id;
        if (hipMemPoolCreate(&d.pool, &pp) == hipSuccess) {
            uint64_t keep = getenv("DIL_POOL_KEEP") ? strtoull(getenv("DIL_POOL_KEEP"), nullptr, 10) : (uint64_t)8 << 30;
            (void)hipMemPoolSetAttribute(d.pool, hipMemPoolAttrReleaseThreshold, &keep);
        } else {
            d.pool = nullptr;        // fall back to the default pool, attributes untouched
        }
        (void)hipGetLastError();
    }
    return 0;
}

// tear one device down (caller holds d.mu; `d.id` is made current for the frees and restored by the caller)
void destroy_device(Device& d)
{
    ::mailbox_destroy(d);
    d.arenas.clear();
    d.aux.destroy();
    if (d.hp.ready) {
        for (int i = 0; i < HOST_STREAMS; i++) {
            if (d.hp.dev[i]) (void)hipFree(d.hp.dev[i]);
            if (d.hp.host[i]) (void)hipHostFree(d.hp.host[i]);
            if (d.hp.stream[i]) (void)hipStreamDestroy(d.hp.stream[i]);
            if (d.hp.up_done[i]) (void)hipEventDestroy(d.hp.up_done[i]);
            if (d.hp.dn_done[i]) (void)hipEventDestroy(d.hp.dn_done[i]);
        }
        d.hp = HostPipe{};
    }
    if (d.d_tables) (void)hipFree(d.d_tables);
    if (d.scratch) (void)hipFree(d.scratch);
    if (d.stage) (void)hipHostFree(d.stage);
    d.stage = nullptr;
    d.stage_bytes = 0;
    if (d.pool) (void)hipMemPoolDestroy(d.pool);
    d.d_tables = nullptr;
    d.scratch = nullptr;
    d.scratch_bytes = 0;
    d.pool = nullptr;
}
}  // namespace

std::atomic<int>* option_slot(const char* name)
{
    int n;
    const OptName* tab = option_table(&n);
    for (int i = 0; i < n; i++)
        if (name && strcmp(name, tab[i].name) == 0) return tab[i].slot;
    return nullptr;
}

int current(Device** out)
{
    int id = 0;
    DIL_TRY(hipGetDevice(&id));
    if (id < 0 || id >= MAX_DEVICES) return (int)hipErrorInvalidDevice;
    Device& d = g_dev[id];
    if (!d.ready.load(std::memory_order_acquire)) {
        std::call_once(g_env_once, read_env);
        std::lock_guard<std::mutex> lk(d.mu);
        if (!d.ready.load(std::memory_order_relaxed)) {
            const int rc = init_device(d, id);
            if (rc) {
                destroy_device(d);
                return rc;
            }
            d.ready.store(true, std::memory_order_release);
        }
    }
    *out = &d;
    return 0;
}

dil::Tables Device::tables() const
{
    dil::Tables t;
    t.fwd = d_tables;
    t.inv = d_tables + 2048;
    t.inv_pipe = d_tables + 4096;
    t.device = id;
    t.num_cus = num_cus;
    auto pos = [](int v, int dflt) { return v > 0 ? v : dflt; };
    t.ntt_blocks_per_cu = pos(cfg.ntt_blocks_per_cu.load(std::memory_order_relaxed), 8);
    t.wpi_blocks_per_cu = pos(cfg.wpi_blocks_per_cu.load(std::memory_order_relaxed), 8);
    t.fused_wgs_per_cu = pos(cfg.fused_wgs_per_cu.load(std::memory_order_relaxed), 4);
    t.fused_mode = cfg.fused_mode.load(std::memory_order_relaxed);
    return t;
}

bool AuxStream::ensure()
{
    if (s) return true;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { s = nullptr; return false; }
    const bool ok = hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess &&
                    hipEventCreateWithFlags(&join, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        destroy();
        return false;
    }
    return true;
}
void AuxStream::destroy()
{
    if (fork) (void)hipEventDestroy(fork);
    if (join) (void)hipEventDestroy(join);
    if (s) (void)hipStreamDestroy(s);
    s = nullptr;
    fork = join = nullptr;
}
}  // namespace rt
}  // namespace dil

namespace {
using dil::rt::S;
using dil::rt::Device;
using dil::rt::HOST_STREAMS;


// ---- twiddles: zeta^brv8(k), zeta = 1753 (consts.cpp:64-97; zetas.txt holds them mod q) ----
unsigned brv8(unsigned x)
{
    unsigned r = 0;
    for (int i = 0; i < 8; i++) r |= ((x >> i) & 1u) << (7 - i);
    return r;
}
int64_t powmod(int64_t b, unsigned e)
{
    int64_t r = 1;
    for (b %= Q; e; e >>= 1, b = b * b % Q)
        if (e & 1) r = r * b % Q;
    return r;
}
void canonical_zetas(uint32_t z[256])
{
    z[0] = 0;
    for (unsigned k = 1; k < 256; k++) z[k] = (uint32_t)powmod(1753, brv8(k));
}
// Montgomery form of a table constant w: wt = centred(w * 2^32 mod q), wq = wt * q^-1 mod 2^32
constexpr uint32_t QINV = 58728449u;
inline void mont_const(uint32_t w, uint32_t* out)
{
    int64_t wt = (int64_t)(((unsigned __int128)w << 32) % (uint64_t)Q);
    if (wt > (Q - 1) / 2) wt -= Q;
    out[0] = (uint32_t)(int32_t)wt;
    out[1] = (uint32_t)(int32_t)wt * QINV;
}

// forward: pass p, lane -> k1 = 4^p + (lane >> (6 - 2p)); entry {z[k1], z[2k1], z[2k1+1]} x (wt, wq), 0, 0
//          (ref_ntt2x2.cpp:50-55 == twiddle_resolver.v:106-130 under the lane layout of ntt_core.hpp)
// inverse: pass p, block t = lane >> 2p (0 in the last pass), base = 256 >> 2p:
//          ka = base-1-2t, ka-1, kb = base/2-1-t, each negated (ref_ntt2x2.cpp:113-118 ==
//          twiddle_resolver.v:87-105); last pass: wb *= f and f rides in slots 6,7, with
//          f = 256^-1 (standalone) or 2^32 * 256^-1 (pipelines, see kernels.hpp)
// slot i (0..7) of (pass p, lane) in the [pass][half][lane][4] layout the kernels read
inline uint32_t* slot(uint32_t* tab, int p, int lane, int i) { return tab + p * 512 + (i >> 2) * 256 + lane * 4 + (i & 3); }

void build_tables(uint32_t* fwd, uint32_t* inv, uint32_t* inv_pipe)
{
    uint32_t z[256];
    canonical_zetas(z);
    const uint64_t f_std = 8347681u;
    const uint64_t f_pipe = f_std * ((1ull << 32) % (uint64_t)Q) % (uint64_t)Q;
    for (int p = 0; p < 4; p++) {
        for (int lane = 0; lane < 64; lane++) {
            const unsigned k1 = (1u << (2 * p)) + ((unsigned)lane >> (6 - 2 * p));
            const uint32_t wf[3] = {z[k1], z[2 * k1], z[2 * k1 + 1]};
            uint32_t pr[2];
            for (int i = 0; i < 3; i++) {
                mont_const(wf[i], pr);
                *slot(fwd, p, lane, 2 * i) = pr[0];
                *slot(fwd, p, lane, 2 * i + 1) = pr[1];
            }
            *slot(fwd, p, lane, 6) = *slot(fwd, p, lane, 7) = 0;

            const unsigned t = (p < 3) ? ((unsigned)lane >> (2 * p)) : 0u;
            const unsigned base = 256u >> (2 * p);
            const unsigned ka = base - 1 - 2 * t, kb = (base >> 1) - 1 - t;
            for (int flavour = 0; flavour < 2; flavour++) {
                uint32_t* d = flavour ? inv_pipe : inv;
                const uint64_t f = flavour ? f_pipe : f_std;
                uint64_t wi[4] = {(uint64_t)((Q - z[ka]) % Q), (uint64_t)((Q - z[ka - 1]) % Q),
                                  (uint64_t)((Q - z[kb]) % Q), f};
                if (p == 3) wi[2] = wi[2] * f % (uint64_t)Q;
                for (int i = 0; i < 4; i++) {
                    mont_const((uint32_t)wi[i], pr);
                    *slot(d, p, lane, 2 * i) = pr[0];
                    *slot(d, p, lane, 2 * i + 1) = pr[1];
                }
            }
        }
    }
}





// ---- host mailbox (kernels.hpp Mailbox): batch-of-one calls without a launch ----------------------------------
constexpr int MB_FALLBACK = -1;          // not served here (off, busy, broken): the caller takes the launch path
inline uint32_t mb_read(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
double now_us()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
int mailbox_start(Device& d, const dil::Tables& T)          // (re)launch the resident wave; d.mbox.mu held
{
    dil::rt::MailboxHost& m = d.mbox;
    __atomic_store_n(&m.host->state, (uint32_t)dil::MB_ALIVE, __ATOMIC_RELEASE);
    const uint64_t ticks = (uint64_t)std::max(1, dil::rt::cfg.mailbox_idle_us.load(std::memory_order_relaxed)) * 100;   // 100 MHz
    const uint64_t resident = (uint64_t)std::max(1, dil::rt::cfg.mailbox_resident_us.load(std::memory_order_relaxed)) * 100;
    m.launches++;
    const int rc = (int)dil::launch_mailbox(m.dev, __atomic_load_n(&m.host->done_seq, __ATOMIC_ACQUIRE), ticks, resident, T, m.stream);
    if (rc) __atomic_store_n(&m.host->state, (uint32_t)dil::MB_DEAD, __ATOMIC_RELEASE);     // no wave was launched: nothing is resident
    return rc;
}
// one request: in0 (and in1) are copied into the mailbox, the wave is woken (or launched), `out` receives the 1 KiB result
int mailbox_call(int op, int mapping, const int32_t* in0, const int32_t* in1, int32_t* out)
{
    if (!dil::rt::cfg.host_mailbox.load(std::memory_order_relaxed)) return MB_FALLBACK;
    DIL_ENTER(d, T);
    dil::rt::MailboxHost& m = d.mbox;
    std::unique_lock<std::mutex> lk(m.mu, std::try_to_lock);
    if (!lk.owns_lock() || m.broken) return MB_FALLBACK;
    if (!m.host) {
        void* p = nullptr;
        if (hipHostMalloc(&p, sizeof(dil::Mailbox), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
            hipHostGetDevicePointer(reinterpret_cast<void**>(&m.dev), p, 0) != hipSuccess ||
            hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            if (p) (void)hipHostFree(p);
            m.broken = true;
            return MB_FALLBACK;
        }
        m.host = static_cast<dil::Mailbox*>(p);
        memset(m.host, 0, sizeof(dil::Mailbox));
    }
    dil::Mailbox* mb = m.host;
    memcpy(mb->in0, in0, 1024);
    if (in1) memcpy(mb->in1, in1, 1024);
    const uint32_t seq = dil::mb_header(++m.seq, (uint32_t)op, (uint32_t)mapping);     // one word: never torn against the wave's poll
    __atomic_store_n(&mb->req_seq, seq, __ATOMIC_RELEASE);
    __atomic_thread_fence(__ATOMIC_SEQ_CST);                  // the request is out before `state` is read (see mailbox_kernel's retirement)
    m.calls++;
    const double t0 = now_us();
    for (uint64_t spins = 0;; spins++) {
        if (mb_read(&mb->done_seq) == seq) break;
        if (mb_read(&mb->state) == (uint32_t)dil::MB_DEAD) {
            if (mb_read(&mb->done_seq) == seq) break;          // served just before it retired
            const int rc = mailbox_start(d, T);
            if (rc) {
                m.broken = true;
                return rc;
            }
        }
        if ((spins & 1023) == 1023 && now_us() - t0 > 2e6) {   // 2 s without an answer (e.g. a saturated GPU that cannot schedule the wave):
            m.broken = true;                                   // never again -- and THIS call takes the launch path too, which still works
            return MB_FALLBACK;                                // (a late device write to mb->out lands in the mailbox, not in the caller's buffer)
        }
        __builtin_ia32_pause();
    }
    memcpy(out, mb->out, 1024);
    return 0;
}
// Tear the mailbox down -- but never under a wave that may still be resident.  The wave polls and writes the mapped page (1 KiB of
// result + two words); freeing that page or destroying its stream while it runs is a device-side use-after-free (or a hang inside
// hipHostFree).  So: ask the wave to leave, wait a BOUNDED time for the `state` word it writes last (MB_DEAD; the page is host memory, the
// wait needs no runtime call), and only then synchronise the stream and free.  A wave that does not answer within the bound -- the state
// `broken` records, or a GPU too busy to schedule it -- keeps its page and its stream: both are leaked on purpose (one page, one stream, once
// per process).
void mailbox_destroy(Device& d)
{
    dil::rt::MailboxHost& m = d.mbox;
    std::lock_guard<std::mutex> lk(m.mu);
    if (!m.host) return;
    bool dead = mb_read(&m.host->state) == (uint32_t)dil::MB_DEAD || m.launches == 0;
    if (!dead) {
        __atomic_store_n(&m.host->req_seq, dil::mb_header(++m.seq, (uint32_t)dil::MB_QUIT, 0), __ATOMIC_RELEASE);
        const double bound_us = 2.0 * std::max(1, dil::rt::cfg.mailbox_resident_us.load(std::memory_order_relaxed)) + 2e5;   // the wave's own residency cap, twice, + 0.2 s
        const double t0 = now_us();
        for (uint64_t spins = 0; !dead; spins++) {
            dead = mb_read(&m.host->state) == (uint32_t)dil::MB_DEAD;
            if (!dead && (spins & 255) == 255 && now_us() - t0 > bound_us) break;
            __builtin_ia32_pause();
        }
    }
    if (dead) {
        (void)hipStreamSynchronize(m.stream);        // the kernel's last instructions after its MB_DEAD store
        (void)hipStreamDestroy(m.stream);
        (void)hipHostFree(m.host);
    } else {
        m.leaked++;                                  // (dil_mailbox_stats reports it)
    }
    m.host = m.dev = nullptr;
    m.stream = nullptr;
    m.broken = false;
    m.launches = 0;
}

int ensure_scratch(Device& d, size_t bytes)
{
    if (bytes <= d.scratch_bytes) return 0;
    if (d.scratch) {
        DIL_TRY(hipFree(d.scratch));
        d.scratch = nullptr;
        d.scratch_bytes = 0;
    }
    DIL_TRY(hipMalloc(&d.scratch, bytes));
    d.scratch_bytes = bytes;
    return 0;
}

// ---- host-pointer entry points: the reference's callers hold HOST buffers (reference_code/ref_ntt.h:30-36) ----------------------------
// RULE (round 6): the library never makes the runtime page-lock, and never page-locks or releases itself, a page of the CALLER's memory.
// On this platform page-locking paged memory is not a counted reference but a per-range ATTRIBUTE of the driver's shared-virtual-memory
// ranges (GPU access in place, at the CPU address), set by whoever locks and cleared by whoever unlocks -- the runtime for every large copy from /
// to pageable memory (it keeps such locks in small per-stream caches and releases them on eviction), hipHostRegister / hipHostUnregister for
// an explicit lock.  Locks and releases of NEIGHBOURING heap ranges by different owners interfere: the round-5 test suite died in ~45 % of
// its runs of "Memory access fault by GPU ... on address <a heap address>" (reason "Unknown" or "Write access to a read-only page") inside a
// plain torch copy from / to a numpy array, each time within milliseconds-to-seconds of this library's own lock traffic on heap ranges next to
// it -- the runtime's implicit locks of rounds 4-5 (pageable caller pointers given to hipMemcpyAsync on the library's private streams) and,
// just the same, explicit hipHostRegister / hipHostUnregister around the call (tried and measured in round 6: 2 of 7 runs died).  The record:
// profiles/r06_suite_crash_rootcause.txt.  So every byte between a pageable caller buffer and the device goes through the library's OWN
// page-locked staging buffers (hipHostMalloc, made once, never attached to caller memory):
//      caller buffer --memcpy (calling thread + pool threads)--> page-locked slot --DMA--> device slot --kernel--> --DMA--> slot --memcpy--> caller
// in a ring of HOST_RING slots on as many streams, so that the memcpy of slice k + 1 runs under the DMA and the kernel of slice k.  The
// memcpy is the bound (two or three threads reach the link's rate, option host_copy_threads).  A buffer the CALLER page-locked (hipHostMalloc,
// its own hipHostRegister, torch pin_memory) needs no lock from anyone and is DMA'd in place: one stream per direction from 64 MiB (the one
// pattern in which the link runs duplex, profiles/r05t_pcie_duplex.txt), chunks round-robin over the streams below.
// Locking: these entry points share the device's staging slots, so they are serialised by the device's `host_mu` -- a lock of their own;
// initialisation (Device::mu) and every *_dev entry point are never blocked by it.
constexpr size_t STAGE_POLYS = 4096;                       // polynomials (KiB) per slice of a pageable transform call = size of a staging slot
constexpr int HOST_RING = 3;                               // staging slots (and streams) a pageable call goes round
static size_t host_chunk_polys()
{
    return (size_t)std::min(std::max(dil::rt::cfg.host_chunk.load(std::memory_order_relaxed), 64), 1 << 20);
}
static bool is_page_locked(const void* h, size_t bytes)          // both ends of the range: a partly registered buffer is pageable to us
{
    if (!h || !bytes) return true;
    const char* ends[2] = {static_cast<const char*>(h), static_cast<const char*>(h) + (bytes - 1)};
    for (const char* p : ends) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) != hipSuccess) {
            (void)hipGetLastError();                  // an ordinary malloc'ed pointer is "invalid value" to the runtime: pageable
            return false;                             // (that clears the runtime's per-thread last-error record, which this query itself has just set)
        }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}
static int host_stream_count() { return std::min(std::max(dil::rt::cfg.host_streams.load(std::memory_order_relaxed), 1), HOST_STREAMS); }

// memcpy on the calling thread + pool threads that never call the HIP runtime (copy_pool.hpp), option host_copy_threads
struct HostCopy {
    dil::CopyPool pool;
    void copy(void* dst, const void* src, size_t n) { pool.copy(dst, src, n, dil::rt::cfg.host_copy_threads.load(std::memory_order_relaxed)); }
};
HostCopy g_copy;

// the library's own page-locked staging buffer for single pieces (per device, under host_mu; grown on demand, kept)
int ensure_stage(Device& d, size_t bytes)
{
    if (bytes <= d.stage_bytes) return 0;
    if (d.stage) {
        DIL_TRY(hipHostFree(d.stage));
        d.stage = nullptr;
        d.stage_bytes = 0;
    }
    DIL_TRY(hipHostMalloc(&d.stage, bytes, hipHostMallocDefault));
    d.stage_bytes = bytes;
    return 0;
}

// streams + events once; device staging buffers only for the `nbuf` a call goes round, each grown to what the call needs -- and given back
// when a run of 16 later calls needs less than a quarter of it, or not that buffer at all (one large dil_verify_core_host call must not hold
// 8 x 64 MiB for the life of the process; calls of two sizes taking turns must not free and allocate every time).  host_slots: the same
// number of page-locked HOST slots of that size (the ring of a pageable call).
int ensure_pipe(Device& d, size_t bytes_per_buffer, int nbuf, bool host_slots = false)
{
    dil::rt::HostPipe& hp = d.hp;
    if (!hp.ready) {
        for (int i = 0; i < HOST_STREAMS; i++) DIL_TRY(hipStreamCreateWithFlags(&hp.stream[i], hipStreamNonBlocking));
        for (int i = 0; i < HOST_STREAMS; i++) {
            DIL_TRY(hipEventCreateWithFlags(&hp.up_done[i], hipEventDisableTiming));
            DIL_TRY(hipEventCreateWithFlags(&hp.dn_done[i], hipEventDisableTiming));
        }
        hp.ready = true;
    }
    for (int i = 0; i < HOST_STREAMS; i++) {
        const bool used = i < nbuf;
        const bool regrow = used && hp.dev_bytes[i] < bytes_per_buffer;
        const bool big = hp.dev[i] && hp.dev_bytes[i] > ((size_t)16 << 20) && (!used || hp.dev_bytes[i] > 4 * bytes_per_buffer);
        hp.oversized[i] = big ? hp.oversized[i] + 1 : 0;
        const bool shrink = hp.oversized[i] >= 16;
        if ((regrow || shrink) && hp.dev[i]) {
            DIL_TRY(hipFree(hp.dev[i]));            // (every stream of the pipe is idle between calls: each call ends in their synchronisation)
            hp.dev[i] = nullptr;
            hp.dev_bytes[i] = 0;
            hp.oversized[i] = 0;
        }
        if (used && !hp.dev[i]) {
            DIL_TRY(hipMalloc(reinterpret_cast<void**>(&hp.dev[i]), bytes_per_buffer));
            hp.dev_bytes[i] = bytes_per_buffer;
        }
        if (host_slots && used && hp.host_bytes[i] < bytes_per_buffer) {
            if (hp.host[i]) DIL_TRY(hipHostFree(hp.host[i]));
            hp.host[i] = nullptr;
            hp.host_bytes[i] = 0;
            DIL_TRY(hipHostMalloc(reinterpret_cast<void**>(&hp.host[i]), bytes_per_buffer, hipHostMallocDefault));
            hp.host_bytes[i] = bytes_per_buffer;
        }
    }
    return 0;
}

// The ring of a call on PAGEABLE caller memory: slice k lives in slot k % nb (a page-locked host buffer, a device buffer, a stream).
//   fill(k, host_slot) -> bytes to upload from the start of the slot         (memcpy from the caller's arrays, calling thread + pool)
//   launch(k, device_slot, stream) -> status                                  (the kernels of the slice)
//   down(k) -> {offset, bytes} of the slot that come back                     drain(k, host_slot)   (memcpy into the caller's arrays)
// The memcpy into slot b of slice k + nb waits for slice k's download (a stream synchronisation: the slot's stream carries nothing else).
template <class Fill, class Launch, class Down, class Drain>
int staged_ring(Device& d, size_t nslices, size_t slot_bytes, Fill&& fill, Launch&& launch, Down&& down, Drain&& drain)
{
    const int nb = (int)std::min<size_t>(nslices, HOST_RING);
    int rc = ensure_pipe(d, slot_bytes, nb, true);
    if (rc) return rc;
    dil::rt::HostPipe& hp = d.hp;
    int err = 0;
    size_t k = 0;
    auto finish = [&](size_t j) {
        const int b = (int)(j % nb);
        const hipError_t e = hipStreamSynchronize(hp.stream[b]);
        if (!err && e != hipSuccess) err = (int)e;
        if (!err) drain(j, reinterpret_cast<const char*>(hp.host[b]));
    };
    for (; k < nslices && !err; k++) {
        const int b = (int)(k % nb);
        if (k >= (size_t)nb) finish(k - nb);
        if (err) break;
        char* hs = reinterpret_cast<char*>(hp.host[b]);
        const size_t up = fill(k, hs);
        err = (int)hipMemcpyAsync(hp.dev[b], hs, up, hipMemcpyHostToDevice, hp.stream[b]);
        if (!err) err = launch(k, reinterpret_cast<char*>(hp.dev[b]), hp.stream[b]);
        const std::pair<size_t, size_t> dn = down(k);
        if (!err) err = (int)hipMemcpyAsync(hs + dn.first, hp.dev[b] + dn.first, dn.second, hipMemcpyDeviceToHost, hp.stream[b]);
    }
    for (size_t j = k > (size_t)nb ? k - nb : 0; j < k; j++) finish(j);          // (on an error: the streams are drained, nothing more is copied out)
    if (err)
        for (int b = 0; b < nb; b++) (void)hipStreamSynchronize(hp.stream[b]);
    return err;
}

// Which form a host-pointer transform call takes, and in which chunks (profiles/r05t_host_batch_sweep.txt for the thresholds of the
// page-locked pipelines)
enum { HOST_PIPE_ONE_SHOT = 0, HOST_PIPE_ROUND_ROBIN = 1, HOST_PIPE_DUPLEX = 2, HOST_PIPE_STAGED_RING = 3 };
struct HostPlan {
    int pipeline;
    size_t chunk;      // polynomials per chunk
};
static HostPlan host_plan(size_t batch, bool locked)
{
    const size_t opt_chunk = host_chunk_polys();
    if (!locked) {                                   // pageable: slices of at most 4 MiB through the ring of page-locked slots
        const size_t sl = std::min(STAGE_POLYS, opt_chunk);
        return batch <= sl ? HostPlan{HOST_PIPE_ONE_SHOT, batch} : HostPlan{HOST_PIPE_STAGED_RING, sl};
    }
    const bool want_duplex = dil::rt::cfg.host_duplex.load(std::memory_order_relaxed) != 0;
    const bool duplex = want_duplex && batch >= 8 * opt_chunk;
    const size_t chunk = (duplex || !want_duplex) ? opt_chunk : std::min<size_t>(opt_chunk, 1024);
    if (batch <= chunk || batch < std::min<size_t>(8192, 2 * opt_chunk)) return {HOST_PIPE_ONE_SHOT, batch};
    return {duplex ? HOST_PIPE_DUPLEX : HOST_PIPE_ROUND_ROBIN, chunk};
}

template <class F>
int host_inplace(int32_t* h, size_t batch, F&& fn)   // fn(device_ptr, n_polys, tables, stream) -> int
{
    if (batch == 0) return 0;
    DIL_ENTER(d, T);
    std::lock_guard<std::mutex> lk(d.host_mu);
    const bool locked = is_page_locked(h, batch * 1024);
    const HostPlan plan = host_plan(batch, locked);
    const size_t HOST_CHUNK = plan.chunk;
    if (!locked) {
        const size_t nsl = (batch + HOST_CHUNK - 1) / HOST_CHUNK;
        auto cnt = [&](size_t k) { return std::min(HOST_CHUNK, batch - k * HOST_CHUNK); };
        return staged_ring(
            d, nsl, std::min(batch, HOST_CHUNK) * 1024,
            [&](size_t k, char* hs) { g_copy.copy(hs, h + k * HOST_CHUNK * 256, cnt(k) * 1024); return cnt(k) * 1024; },
            [&](size_t k, char* dv, hipStream_t st) { return fn(reinterpret_cast<int32_t*>(dv), cnt(k), T, st); },
            [&](size_t k) { return std::make_pair((size_t)0, cnt(k) * 1024); },
            [&](size_t k, const char* hs) { g_copy.copy(h + k * HOST_CHUNK * 256, hs, cnt(k) * 1024); });
    }
    // the caller's own page-locked memory: DMA in place
    if (plan.pipeline == HOST_PIPE_ONE_SHOT) {
        const size_t bytes = batch * 1024;
        int rc = ensure_scratch(d, bytes);
        if (rc) return rc;
        DIL_TRY(hipMemcpy(d.scratch, h, bytes, hipMemcpyHostToDevice));
        rc = fn(static_cast<int32_t*>(d.scratch), batch, T, (hipStream_t)0);
        if (rc) return rc;
        DIL_TRY(hipStreamSynchronize(nullptr));
        DIL_TRY(hipMemcpy(h, d.scratch, bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    const int NS = host_stream_count();
    int rc = ensure_pipe(d, HOST_CHUNK * 1024, NS);
    if (rc) return rc;
    dil::rt::HostPipe& hp = d.hp;
    int err = 0;
    size_t c = 0;
    if (plan.pipeline == HOST_PIPE_DUPLEX) {
        // The link carries both directions at once only in ONE pattern of those tried (profiles/r05t_pcie_duplex.txt): one stream per
        // direction, chunks of 4 - 16 MiB -- 43 - 47 GB/s each way, where two large copies side by side share 57 GB/s and several streams
        // per direction fall back to ~32.  So: stream 0 carries every upload, stream 1 the kernels and downloads, NS staging buffers go
        // round between them on events.
        hipStream_t up = hp.stream[0], dn = hp.stream[1];
        for (size_t off = 0; off < batch && !err; off += HOST_CHUNK, c++) {
            const int b = (int)(c % NS);
            const size_t n = batch - off < HOST_CHUNK ? batch - off : HOST_CHUNK;
            int32_t* hc = h + off * 256;
            int32_t* dc = reinterpret_cast<int32_t*>(hp.dev[b]);
            if (c >= (size_t)NS) err = (int)hipStreamWaitEvent(up, hp.dn_done[b], 0);        // the buffer's previous chunk has left
            if (!err) err = (int)hipMemcpyAsync(dc, hc, n * 1024, hipMemcpyHostToDevice, up);
            if (!err) err = (int)hipEventRecord(hp.up_done[b], up);
            if (!err) err = (int)hipStreamWaitEvent(dn, hp.up_done[b], 0);
            if (!err) err = fn(dc, n, T, dn);
            if (!err) err = (int)hipMemcpyAsync(hc, dc, n * 1024, hipMemcpyDeviceToHost, dn);
            if (!err) err = (int)hipEventRecord(hp.dn_done[b], dn);
        }
        for (int i = 0; i < 2; i++) {
            const hipError_t e = hipStreamSynchronize(hp.stream[i]);
            if (!err && e != hipSuccess) err = (int)e;
        }
        return err;
    }
    for (size_t off = 0; off < batch && !err; off += HOST_CHUNK, c++) {
        const int s = (int)(c % NS);
        const size_t n = batch - off < HOST_CHUNK ? batch - off : HOST_CHUNK;
        int32_t* hc = h + off * 256;
        int32_t* dc = reinterpret_cast<int32_t*>(hp.dev[s]);
        err = (int)hipMemcpyAsync(dc, hc, n * 1024, hipMemcpyHostToDevice, hp.stream[s]);
        if (!err) err = fn(dc, n, T, hp.stream[s]);
        if (!err) err = (int)hipMemcpyAsync(hc, dc, n * 1024, hipMemcpyDeviceToHost, hp.stream[s]);
    }
    for (int i = 0; i < NS; i++) {
        const hipError_t e = hipStreamSynchronize(hp.stream[i]);
        if (!err && e != hipSuccess) err = (int)e;
    }
    return err;
}

// The verify core from HOST operands (the reference's calling convention for the path: caller-owned host arrays, ref_ntt.h:30-36):
// items in chunks over the streams, each chunk  H2D (A, z, c, t1, h) -> fused kernel -> D2H (w1).  A key shared by the batch goes up once per
// slot.  Pageable operands (any of the six): the chunk's image is assembled in a page-locked slot of the ring (16 MiB slots) and goes up
// as ONE copy; operands the caller page-locked: five copies straight from the arrays, chunks of as many items as fit host_chunk KiB (at least 64 MiB).
int host_verify_core(uint8_t* w1, const int32_t* A, const int32_t* z, const int32_t* c, const int32_t* t1, const uint8_t* h, int level, size_t batch,
                     int shared_pk)
{
    if (batch == 0) return 0;
    if (level != 2 && level != 3 && level != 5) return (int)hipErrorInvalidValue;
    DIL_ENTER(d, T);
    std::lock_guard<std::mutex> lk(d.host_mu);
    const size_t K = level == 2 ? 4 : level == 3 ? 6 : 8, L = level == 2 ? 4 : level == 3 ? 5 : 7;
    const size_t bA = K * L * 1024, bz = L * 1024, bc = 1024, bt = K * 1024, bh = K * 256, bw = K * 256;
    const size_t key_bytes = bA + bt, item_in = bz + bc + bh + (shared_pk ? 0 : key_bytes), item_all = item_in + bw;
    const size_t nk = shared_pk ? 1 : batch;
    const bool locked = is_page_locked(A, nk * bA) && is_page_locked(z, batch * bz) && is_page_locked(c, batch * bc) && is_page_locked(t1, nk * bt) &&
                        is_page_locked(h, batch * bh) && is_page_locked(w1, batch * bw);
    const size_t opt = std::max(host_chunk_polys(), (size_t)64) * 1024;
    const size_t budget = locked ? std::max(opt, (size_t)64 << 20) : std::min(std::max(opt, (size_t)1 << 20), (size_t)16 << 20);      // (upload-dominated: 64 MiB chunks reach 0.88-0.95 of the link)
    const size_t key_in_slot = shared_pk ? key_bytes : 0;
    const size_t per_chunk = std::max<size_t>(1, (budget > key_in_slot ? budget - key_in_slot : 0) / item_all);
    const size_t slot_bytes = std::max(budget, per_chunk * item_all + key_in_slot);
    // slot layout: [A | t1 of the shared key] then per chunk A, t1 (a key per item), z, c, h, w1 -- every block 1 KiB aligned
    struct Lay { size_t oA, ot, oz, oc, oh, ow; };
    auto lay = [&](size_t n) {
        const size_t oq = shared_pk ? key_bytes : n * key_bytes;
        return Lay{0, shared_pk ? bA : n * bA, oq, oq + n * bz, oq + n * (bz + bc), oq + n * (bz + bc) + n * bh};
    };
    auto cnt = [&](size_t k) { return std::min(per_chunk, batch - k * per_chunk); };
    const size_t nch = (batch + per_chunk - 1) / per_chunk;
    if (!locked) {
        bool key_up[HOST_STREAMS] = {};
        const int nb = (int)std::min<size_t>(nch, HOST_RING);
        return staged_ring(
            d, nch, slot_bytes,
            [&](size_t k, char* hs) {
                const size_t n = cnt(k), off = k * per_chunk;
                const Lay o = lay(n);
                const int b = (int)(k % nb);
                if (!shared_pk) {
                    g_copy.copy(hs + o.oA, A + off * (bA / 4), n * bA);
                    g_copy.copy(hs + o.ot, t1 + off * (bt / 4), n * bt);
                } else if (!key_up[b]) {             // the key rides in every slot (host and device side) from its first use on
                    memcpy(hs + o.oA, A, bA);
                    memcpy(hs + o.ot, t1, bt);
                    key_up[b] = true;
                }
                g_copy.copy(hs + o.oz, z + off * (bz / 4), n * bz);
                g_copy.copy(hs + o.oc, c + off * (bc / 4), n * bc);
                g_copy.copy(hs + o.oh, h + off * bh, n * bh);
                return o.ow;                         // (a shared key is uploaded again with each slice: 36 KiB of a 16-MiB slot)
            },
            [&](size_t k, char* dv, hipStream_t st) {
                const size_t n = cnt(k);
                const Lay o = lay(n);
                return (int)dil::launch_verify(level, reinterpret_cast<uint8_t*>(dv + o.ow), reinterpret_cast<int32_t*>(dv + o.oA), reinterpret_cast<int32_t*>(dv + o.oz),
                                               reinterpret_cast<int32_t*>(dv + o.oc), reinterpret_cast<int32_t*>(dv + o.ot), reinterpret_cast<uint8_t*>(dv + o.oh), n,
                                               shared_pk, T, st);
            },
            [&](size_t k) { return std::make_pair(lay(cnt(k)).ow, cnt(k) * bw); },
            [&](size_t k, const char* hs) { g_copy.copy(w1 + k * per_chunk * bw, hs + lay(cnt(k)).ow, cnt(k) * bw); });
    }
    const int NS = host_stream_count();
    int rc = ensure_pipe(d, slot_bytes, NS);
    if (rc) return rc;
    dil::rt::HostPipe& hp = d.hp;
    int err = 0;
    bool key_up[HOST_STREAMS] = {};
    for (size_t ck = 0; ck < nch && !err; ck++) {
        const int s = (int)(ck % NS);
        const size_t n = cnt(ck), off = ck * per_chunk;
        const Lay o = lay(n);
        hipStream_t st = hp.stream[s];
        uint8_t* base = hp.dev[s];
        if (shared_pk) {
            if (!key_up[s]) {                        // once per stream's staging buffer
                err = (int)hipMemcpyAsync(base + o.oA, A, bA, hipMemcpyHostToDevice, st);
                if (!err) err = (int)hipMemcpyAsync(base + o.ot, t1, bt, hipMemcpyHostToDevice, st);
                key_up[s] = true;
            }
        } else {
            err = (int)hipMemcpyAsync(base + o.oA, A + off * (bA / 4), n * bA, hipMemcpyHostToDevice, st);
            if (!err) err = (int)hipMemcpyAsync(base + o.ot, t1 + off * (bt / 4), n * bt, hipMemcpyHostToDevice, st);
        }
        if (!err) err = (int)hipMemcpyAsync(base + o.oz, z + off * (bz / 4), n * bz, hipMemcpyHostToDevice, st);
        if (!err) err = (int)hipMemcpyAsync(base + o.oc, c + off * (bc / 4), n * bc, hipMemcpyHostToDevice, st);
        if (!err) err = (int)hipMemcpyAsync(base + o.oh, h + off * bh, n * bh, hipMemcpyHostToDevice, st);
        if (!err)
            err = (int)dil::launch_verify(level, base + o.ow, reinterpret_cast<int32_t*>(base + o.oA), reinterpret_cast<int32_t*>(base + o.oz),
                                          reinterpret_cast<int32_t*>(base + o.oc), reinterpret_cast<int32_t*>(base + o.ot), base + o.oh, n, shared_pk, T, st);
        if (!err) err = (int)hipMemcpyAsync(w1 + off * bw, base + o.ow, n * bw, hipMemcpyDeviceToHost, st);
    }
    for (int i = 0; i < NS; i++) {
        const hipError_t e = hipStreamSynchronize(hp.stream[i]);
        if (!err && e != hipSuccess) err = (int)e;
    }
    return err;
}

// two-operand host forms (pointwise / bram mul / the fused polynomial product): out <- fn(a, b).  a | b of a slice share one slot, the result
// overwrites a's half.  Pageable operands: the ring of page-locked slots; all three arrays page-locked by the caller: DMA in place, chunks
// round-robin over the streams.
template <class F>
int host_binary(int32_t* out, const int32_t* a, const int32_t* b, size_t batch, F&& fn)      // fn(da, db, n, tables, stream) -> int, result in da
{
    if (batch == 0) return 0;
    DIL_ENTER(d, T);
    std::lock_guard<std::mutex> lk(d.host_mu);
    const bool locked = is_page_locked(a, batch * 1024) && is_page_locked(b, batch * 1024) && is_page_locked(out, batch * 1024);
    const size_t chunk = std::max<size_t>((locked ? host_chunk_polys() : std::min(STAGE_POLYS, host_chunk_polys())) / 2, 32);
    const size_t nch = (batch + chunk - 1) / chunk;
    auto cnt = [&](size_t k) { return std::min(chunk, batch - k * chunk); };
    if (!locked)
        return staged_ring(
            d, nch, 2 * std::min(batch, chunk) * 1024,
            [&](size_t k, char* hs) {
                g_copy.copy(hs, a + k * chunk * 256, cnt(k) * 1024);
                g_copy.copy(hs + cnt(k) * 1024, b + k * chunk * 256, cnt(k) * 1024);
                return 2 * cnt(k) * 1024;
            },
            [&](size_t k, char* dv, hipStream_t st) { return fn(reinterpret_cast<int32_t*>(dv), reinterpret_cast<int32_t*>(dv) + cnt(k) * 256, cnt(k), T, st); },
            [&](size_t k) { return std::make_pair((size_t)0, cnt(k) * 1024); },
            [&](size_t k, const char* hs) { g_copy.copy(out + k * chunk * 256, hs, cnt(k) * 1024); });
    const int NS = host_stream_count();
    int rc = ensure_pipe(d, 2 * std::min(batch, chunk) * 1024, NS);
    if (rc) return rc;
    dil::rt::HostPipe& hp = d.hp;
    int err = 0;
    for (size_t k = 0; k < nch && !err; k++) {
        const int s = (int)(k % NS);
        const size_t n = cnt(k), off = k * chunk;
        int32_t* da = reinterpret_cast<int32_t*>(hp.dev[s]);
        int32_t* db = da + n * 256;
        err = (int)hipMemcpyAsync(da, a + off * 256, n * 1024, hipMemcpyHostToDevice, hp.stream[s]);
        if (!err) err = (int)hipMemcpyAsync(db, b + off * 256, n * 1024, hipMemcpyHostToDevice, hp.stream[s]);
        if (!err) err = fn(da, db, n, T, hp.stream[s]);
        if (!err) err = (int)hipMemcpyAsync(out + off * 256, da, n * 1024, hipMemcpyDeviceToHost, hp.stream[s]);
    }
    for (int i = 0; i < NS; i++) {
        const hipError_t e = hipStreamSynchronize(hp.stream[i]);
        if (!err && e != hipSuccess) err = (int)e;
    }
    return err;
}

}  // namespace

// plain copies between a caller's host buffer and device memory under the same rule (scheme.hip's *_host forms): a pageable buffer in slices
// through the library's page-locked staging buffer, one the caller page-locked by DMA in place
namespace dil {
namespace rt {
static int host_copy(Device& d, void* host, void* dev, size_t bytes, bool up)
{
    if (bytes == 0) return 0;
    std::lock_guard<std::mutex> lk(d.host_mu);
    if (is_page_locked(host, bytes)) return (int)hipMemcpy(up ? dev : host, up ? host : dev, bytes, up ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
    const size_t SL = STAGE_POLYS * 1024;
    const int rc = ensure_stage(d, std::min(bytes, SL));
    if (rc) return rc;
    for (size_t off = 0; off < bytes; off += SL) {
        const size_t n = std::min(SL, bytes - off);
        char* h = static_cast<char*>(host) + off;
        char* g = static_cast<char*>(dev) + off;
        if (up) {
            g_copy.copy(d.stage, h, n);
            DIL_TRY(hipMemcpy(g, d.stage, n, hipMemcpyHostToDevice));
        } else {
            DIL_TRY(hipMemcpy(d.stage, g, n, hipMemcpyDeviceToHost));
            g_copy.copy(h, d.stage, n);
        }
    }
    return 0;
}
int host_upload(Device& d, void* dev, const void* host, size_t bytes) { return host_copy(d, const_cast<void*>(host), dev, bytes, true); }
int host_download(Device& d, void* host, const void* dev, size_t bytes) { return host_copy(d, host, const_cast<void*>(dev), bytes, false); }
}  // namespace rt
}  // namespace dil

extern "C" {

void dil_host_twiddle_tables(uint32_t* fwd, uint32_t* inv, uint32_t* inv_pipe) { build_tables(fwd, inv, inv_pipe); }

void dil_host_zetas(int32_t* zetas)
{
    uint32_t z[256];
    canonical_zetas(z);
    for (int k = 0; k < 256; k++) zetas[k] = (int32_t)(z[k] > (uint32_t)(Q - 1) / 2 ? (int64_t)z[k] - Q : z[k]);
}

int dil_device_count(int* count) { return (int)hipGetDeviceCount(count); }

int dil_num_cus(void)
{
    Device* d = nullptr;
    return dil::rt::current(&d) ? -1 : d->num_cus;
}

const char* dil_error_string(int code)
{
    if (code == DIL_ERR_UNFINISHED) return "signing did not finish within max_attempts";
    if (code == DIL_ERR_RCCL) return "RCCL error (dil_multi_last_error has the text)";
    return hipGetErrorString((hipError_t)code);
}

int dil_init(int device)
{
    if (device >= 0) DIL_TRY(hipSetDevice(device));
    Device* d = nullptr;
    return dil::rt::current(&d);
}

int dil_shutdown(void)
{
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int i = 0; i < dil::rt::MAX_DEVICES; i++) {
        Device& d = dil::rt::g_dev[i];
        std::lock_guard<std::mutex> lk(d.mu);
        if (!d.ready.load()) continue;
        if (hipSetDevice(i) == hipSuccess) dil::rt::destroy_device(d);
        d.ready.store(false);
    }
    if (have_cur) (void)hipSetDevice(cur);
    (void)hipGetLastError();
    return 0;
}

int dil_set_option(const char* name, int value)
{
    std::atomic<int>* slot = dil::rt::option_slot(name);
    if (!slot) return (int)hipErrorInvalidValue;
    std::call_once(dil::rt::g_env_once, dil::rt::read_env);     // an explicit setting wins over the environment
    slot->store(value);
    return 0;
}

int dil_get_option(const char* name, int* value)
{
    std::atomic<int>* slot = dil::rt::option_slot(name);
    if (!slot || !value) return (int)hipErrorInvalidValue;
    std::call_once(dil::rt::g_env_once, dil::rt::read_env);
    *value = slot->load();
    return 0;
}

// ---- transforms ---------------------------------------------------------------------------
int dil_ntt_dev(int32_t* polys, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_ntt(false, dil::LAYOUT_POLY, 0, polys, batch, T, S(stream));
}
int dil_invntt_dev(int32_t* polys, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_ntt(true, dil::LAYOUT_POLY, 0, polys, batch, T, S(stream));
}
int dil_ntt_host(int32_t* polys, size_t batch)
{
    if (batch == 1) {
        const int rc = mailbox_call(dil::MB_FWD, 0, polys, nullptr, polys);
        if (rc != MB_FALLBACK) return rc;
    }
    return host_inplace(polys, batch, [](int32_t* p, size_t n, const dil::Tables& t, hipStream_t st) {
        return (int)dil::launch_ntt(false, dil::LAYOUT_POLY, 0, p, n, t, st);
    });
}
int dil_invntt_host(int32_t* polys, size_t batch)
{
    if (batch == 1) {
        const int rc = mailbox_call(dil::MB_INV, 0, polys, nullptr, polys);
        if (rc != MB_FALLBACK) return rc;
    }
    return host_inplace(polys, batch, [](int32_t* p, size_t n, const dil::Tables& t, hipStream_t st) {
        return (int)dil::launch_ntt(true, dil::LAYOUT_POLY, 0, p, n, t, st);
    });
}

// ---- element-wise ---------------------------------------------------------------------------
int dil_pointwise_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_pointwise(dil::OP_MUL, c, a, b, nullptr, batch, T, S(stream));
}
int dil_pointwise_acc_dev(int32_t* c, const int32_t* acc, const int32_t* a, const int32_t* b, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_pointwise(dil::OP_MAC, c, a, b, acc, batch, T, S(stream));
}
int dil_poly_add_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_pointwise(dil::OP_ADD, c, a, b, nullptr, batch, T, S(stream));
}
int dil_poly_sub_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_pointwise(dil::OP_SUB, c, a, b, nullptr, batch, T, S(stream));
}
int dil_pointwise_host(int32_t* c, const int32_t* a, const int32_t* b, size_t batch)
{
    if (batch == 1) {
        const int rc = mailbox_call(dil::MB_PW_MUL, 0, a, b, c);
        if (rc != MB_FALLBACK) return rc;
    }
    return host_binary(c, a, b, batch, [](int32_t* da, int32_t* db, size_t n, const dil::Tables& t, hipStream_t st) {
        return (int)dil::launch_pointwise(dil::OP_MUL, da, da, db, nullptr, n, t, st);
    });
}

// c = a * b in Z_q[x] / (x^256 + 1): the reference's polymul chain (ntt, ntt, pointwise_barrett, invntt; ntt2x2_test.cpp:109-137) fused
int dil_polymul_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    if (batch && (!c || !a || !b)) return (int)hipErrorInvalidValue;
    return (int)dil::launch_polymul(c, a, b, batch, T, S(stream));
}
int dil_polymul_host(int32_t* c, const int32_t* a, const int32_t* b, size_t batch)
{
    if (batch && (!c || !a || !b)) return (int)hipErrorInvalidValue;
    if (batch == 1) {
        const int rc = mailbox_call(dil::MB_POLYMUL, 0, a, b, c);
        if (rc != MB_FALLBACK) return rc;
    }
    return host_binary(c, a, b, batch, [](int32_t* da, int32_t* db, size_t n, const dil::Tables& t, hipStream_t st) {
        return (int)dil::launch_polymul(da, da, db, n, t, st);
    });
}

// ---- bram (hardware-model API) -----------------------------------------------------------------
static int check_mapping(int m) { return (m < 0 || m > 2) ? (int)hipErrorInvalidValue : 0; }

int dil_bram_fwdntt_dev(int32_t* ram, size_t batch, int mapping, void* stream)
{
    if (int rc = check_mapping(mapping)) return rc;
    DIL_ENTER(d, T);
    return (int)dil::launch_ntt(false, dil::LAYOUT_BRAM, mapping, ram, batch, T, S(stream));
}
int dil_bram_invntt_dev(int32_t* ram, size_t batch, int mapping, void* stream)
{
    if (int rc = check_mapping(mapping)) return rc;
    DIL_ENTER(d, T);
    return (int)dil::launch_ntt(true, dil::LAYOUT_BRAM, mapping, ram, batch, T, S(stream));
}
int dil_bram_mul_dev(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping, void* stream)
{
    if (int rc = check_mapping(mapping)) return rc;
    DIL_ENTER(d, T);
    return (int)dil::launch_bram_mul(ram, mul_ram, batch, mapping, T, S(stream));
}
int dil_bram_fwdntt_host(int32_t* ram, size_t batch, int mapping)
{
    if (int rc = check_mapping(mapping)) return rc;
    if (batch == 1) {
        const int rc = mailbox_call(dil::MB_BRAM_FWD, mapping, ram, nullptr, ram);
        if (rc != MB_FALLBACK) return rc;
    }
    return host_inplace(ram, batch, [mapping](int32_t* p, size_t n, const dil::Tables& t, hipStream_t st) {
        return (int)dil::launch_ntt(false, dil::LAYOUT_BRAM, mapping, p, n, t, st);
    });
}
int dil_bram_invntt_host(int32_t* ram, size_t batch, int mapping)
{
    if (int rc = check_mapping(mapping)) return rc;
    if (batch == 1) {
        const int rc = mailbox_call(dil::MB_BRAM_INV, mapping, ram, nullptr, ram);
        if (rc != MB_FALLBACK) return rc;
    }
    return host_inplace(ram, batch, [mapping](int32_t* p, size_t n, const dil::Tables& t, hipStream_t st) {
        return (int)dil::launch_ntt(true, dil::LAYOUT_BRAM, mapping, p, n, t, st);
    });
}
int dil_bram_mul_host(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping)
{
    if (int rc = check_mapping(mapping)) return rc;
    if (batch == 1) {
        const int rc = mailbox_call(dil::MB_BRAM_MUL, mapping, ram, mul_ram, ram);
        if (rc != MB_FALLBACK) return rc;
    }
    return host_binary(ram, ram, mul_ram, batch, [mapping](int32_t* da, int32_t* db, size_t n, const dil::Tables& t, hipStream_t st) {
        return (int)dil::launch_bram_mul(da, db, n, mapping, t, st);
    });
}

int dil_clock_probe_dev(uint64_t* out4, unsigned spin_us, void* stream)
{
    DIL_ENTER(d, T);
    if (!out4) return (int)hipErrorInvalidValue;
    return (int)dil::launch_clock_probe(out4, (uint64_t)spin_us * 100, S(stream));
}

int dil_host_plan(size_t batch, int page_locked, int* pipeline, size_t* chunk_polys)
{
    const HostPlan plan = host_plan(batch, page_locked != 0);
    if (pipeline) *pipeline = plan.pipeline;
    if (chunk_polys) *chunk_polys = plan.chunk;
    return 0;
}

int dil_mailbox_stats(uint64_t* calls, uint64_t* launches, int* alive)
{
    DIL_ENTER(d, T);
    std::lock_guard<std::mutex> lk(d.mbox.mu);
    if (calls) *calls = d.mbox.calls;
    if (launches) *launches = d.mbox.launches;
    if (alive) *alive = d.mbox.host ? (int)mb_read(&d.mbox.host->state) : 0;
    return d.mbox.broken ? (int)hipErrorLaunchTimeOut : 0;
}

int dil_ntt_traffic_dev(int32_t* polys, size_t batch, int inverse, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_ntt_traffic(inverse != 0, polys, batch, T, S(stream));
}

// ---- fused pipelines ---------------------------------------------------------------------------
int dil_matvec_dev(int32_t* w, const int32_t* A, const int32_t* y, int level, size_t batch, int shared_A, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_matvec(level, dil::OUT_W, w, nullptr, nullptr, A, y, batch, shared_A, T, S(stream));
}
int dil_verify_core_dev(uint8_t* w1, const int32_t* A, const int32_t* z, const int32_t* c, const int32_t* t1,
                        const uint8_t* h, int level, size_t batch, int shared_pk, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_verify(level, w1, A, z, c, t1, h, batch, shared_pk, T, S(stream));
}
int dil_verify_core_host(uint8_t* w1, const int32_t* A, const int32_t* z, const int32_t* c, const int32_t* t1, const uint8_t* h, int level,
                         size_t batch, int shared_pk)
{
    return host_verify_core(w1, A, z, c, t1, h, level, batch, shared_pk);
}
int dil_sign_phase1_dev(uint8_t* w1, int32_t* w0, const int32_t* A, const int32_t* y, int level, size_t batch,
                        int shared_key, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_matvec(level, dil::OUT_W1W0, nullptr, w1, w0, A, y, batch, shared_key, T, S(stream));
}
int dil_sign_phase2_dev(int32_t* z, uint8_t* h, int32_t* flags, const int32_t* c, const int32_t* y, const int32_t* w0,
                        const uint8_t* w1, const int32_t* s1hat, const int32_t* s2hat, const int32_t* t0hat, int level,
                        size_t batch, int shared_key, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_sign2(level, z, h, flags, c, y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, T, S(stream));
}

int dil_sign_phase2_skey_dev(int32_t* z, uint8_t* h, int32_t* flags, const int32_t* c, const int32_t* y, int32_t* w0,
                             const uint8_t* w1, const int32_t* s1hat, const int32_t* s2hat, const int32_t* t0hat, int level,
                             size_t batch, int shared_key, int early_exit, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_sign2(level, z, h, flags, c, y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, T, S(stream), dil::KeyMap(),
                                  early_exit ? w0 : nullptr, dil::Y_I32, /*small_key=*/true);
}

int dil_sign_phase2_early_dev(int32_t* z, uint8_t* h, int32_t* flags, const int32_t* c, const int32_t* y, int32_t* w0,
                              const uint8_t* w1, const int32_t* s1hat, const int32_t* s2hat, const int32_t* t0hat, int level,
                              size_t batch, int shared_key, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_sign2(level, z, h, flags, c, y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, T, S(stream), dil::KeyMap(), w0);
}

int dil_launch_info(const char* family, int* grid, int* items_per_block, size_t* items, size_t* launches)
{
    if (!family) return (int)hipErrorInvalidValue;
    int n;
    dil::LaunchRecord* tab = dil::launch_records(&n);
    for (int i = 0; i < n; i++)
        if (!strcmp(tab[i].family, family)) {
            if (grid) *grid = (int)tab[i].grid.load(std::memory_order_relaxed);
            if (items_per_block) *items_per_block = (int)tab[i].items_per_block.load(std::memory_order_relaxed);
            if (items) *items = (size_t)tab[i].items.load(std::memory_order_relaxed);
            if (launches) *launches = (size_t)tab[i].launches.load(std::memory_order_relaxed);
            return 0;
        }
    return (int)hipErrorInvalidValue;
}

// ---- row N1: samplers ---------------------------------------------------------------------------
int dil_shake256_dev(uint8_t* out, size_t out_bytes, const uint8_t* in, size_t in_bytes, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in)) & 7) return (int)hipErrorInvalidValue;
    return (int)dil::launch_shake256(reinterpret_cast<uint64_t*>(out), (int)out_bytes, reinterpret_cast<const uint64_t*>(in),
                                     (int)in_bytes, batch, S(stream));
}
int dil_expand_a_dev(int32_t* A, const uint8_t* rho, int level, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_expand_a(A, rho, 32, level, batch, S(stream));
}
int dil_expand_mask_dev(int32_t* y, const uint8_t* rhoprime, const uint32_t* kappa, int level, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_expand_mask(y, rhoprime, kappa, level, batch, S(stream));
}
int dil_sample_in_ball_dev(int32_t* c, const uint8_t* ctilde, int level, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_sample_in_ball(c, ctilde, level, batch, S(stream));
}
int dil_challenge_dev(uint8_t* ctilde, int32_t* c, const uint8_t* mu, const uint8_t* w1_packed, int level, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    if ((reinterpret_cast<uintptr_t>(ctilde) | reinterpret_cast<uintptr_t>(mu) | reinterpret_cast<uintptr_t>(w1_packed)) & 7)
        return (int)hipErrorInvalidValue;
    return (int)dil::launch_challenge_sample(ctilde, c, mu, w1_packed, level, batch, S(stream));
}
int dil_pack_w1_dev(uint8_t* out, const uint8_t* w1, int level, size_t batch, void* stream)
{
    DIL_ENTER(d, T);
    return (int)dil::launch_pack_w1(out, w1, level, batch, T, S(stream));
}

// ---- events --------------------------------------------------------------------------------------
int dil_event_create(void** ev)
{
    hipEvent_t e;
    DIL_TRY(hipEventCreate(&e));
    *ev = e;
    return 0;
}
int dil_event_destroy(void* ev) { return (int)hipEventDestroy(static_cast<hipEvent_t>(ev)); }
int dil_event_record(void* ev, void* stream) { return (int)hipEventRecord(static_cast<hipEvent_t>(ev), S(stream)); }
int dil_event_elapsed_ms(float* ms, void* start, void* stop)
{
    DIL_TRY(hipEventSynchronize(static_cast<hipEvent_t>(stop)));
    return (int)hipEventElapsedTime(ms, static_cast<hipEvent_t>(start), static_cast<hipEvent_t>(stop));
}
int dil_stream_sync(void* stream) { return (int)hipStreamSynchronize(S(stream)); }

}  // extern "C"
