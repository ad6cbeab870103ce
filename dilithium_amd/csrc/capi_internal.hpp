// capi_internal.hpp -- what the translation units behind include/dil256.h share: the process-wide options, the
// PER-DEVICE runtime state (one `Device` per HIP device, created on first use by whichever thread has that device
// current), lazy initialisation and error propagation.  There is no single "the" device: a C++ host that drives
// eight GPUs from one process (one thread per GPU, or one thread switching with hipSetDevice) gets eight independent
// Device records -- twiddle tables, scratch, arenas, helper stream -- and never shares a lock between them.
#pragma once
#include "../../include/dil256.h"
#include "kernels.hpp"

#include <atomic>
#include <mutex>

#define DIL_TRY(expr)                          \
    do {                                       \
        hipError_t e__ = (expr);               \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

namespace dil {
namespace rt {

constexpr int MAX_DEVICES = 32;

// Process-wide tunables (dil_set_option / environment at first use); read at call time, never cached in a launch.
struct Config {
    std::atomic<int> fused_mode{0};        // DIL_FUSED_MODE: 0 auto (by batch), 1 workgroup-per-item, 2 wave-per-item
    std::atomic<int> ntt_blocks_per_cu{8}; // DIL_NTT_BPC
    std::atomic<int> wpi_blocks_per_cu{8}; // DIL_WPI_BPC
    std::atomic<int> fused_wgs_per_cu{4};  // DIL_FUSED_WGPC
    std::atomic<int> sign_early{1};        // DIL_SIGN_EARLY: 0 = the signing loop evaluates every check of every attempt
    std::atomic<int> sign_skip{3};         // DIL_SIGN_SKIP: bit 0 = phase 2 of a speculative round drops the attempts behind an accepted one, bit 1 = its waves draw entries from a queue
    std::atomic<int> sign_waste{6144};     // DIL_SIGN_WASTE: speculative entries a round may expect to waste
    std::atomic<int> sign_wake{1};         // DIL_SIGN_WAKE: how a signing round's pending count reaches the host: 1 = posted into mapped host words by the collect kernel (polled), 0 = copy + event
    std::atomic<int> sign_cap{0};          // DIL_SIGN_CAP: entries in flight per signing round (0 = default 16384)
    std::atomic<int> aux_overlap{1};       // DIL_AUX_OVERLAP: 0 = composite calls never use the helper stream
    std::atomic<int> zeroize{0};           // DIL_ZEROIZE: 1 = signing / keygen clear their device scratch before returning
    std::atomic<int> fuse_wire{1};         // DIL_FUSE_WIRE: 0 = wire-format verify runs the unfused (codec + core) sequence
    std::atomic<int> fuse_sib{1};          // DIL_FUSE_SIB: wire-format verification with a key per item samples c inside the fused kernel (bit 0: when A is already expanded --
                                           // dil_verify_wire_core_dev, dil_verify_sig_expanded_dev; bit 1: beside ExpandA in dil_verify_sig_dev too); 0: sample_in_ball_bits_kernel in front
    std::atomic<int> fuse_keygen{1};       // DIL_FUSE_KEYGEN: 0 = keygen's mat-vec, Power2Round and t1 / t0 packing as separate kernels
    std::atomic<int> a24{1};               // DIL_A24: 0 = the composite calls keep per-item matrices as int32 in HBM (1: 24-bit packed)
    std::atomic<int> fuse_challenge{1};    // DIL_FUSE_CHALLENGE: 1 = the signing loop hashes c~ and samples c in ONE launch (0: challenge hash, then SampleInBall)
    std::atomic<int> host_mailbox{0};      // DIL_HOST_MAILBOX: 1 = the *_host entry points serve batch == 1 through the resident mailbox wave
                                           // (libdil256_ref.so turns it on: its ntt() / invntt() / ... are batch-of-one calls)
    std::atomic<int> mailbox_idle_us{200}; // DIL_MAILBOX_IDLE_US: the mailbox wave retires after this long without a request
    std::atomic<int> mailbox_resident_us{20000};   // DIL_MAILBOX_RESIDENT_US: ... and after this long in all, busy or not (a device-wide sync of another thread waits at most this long)
    std::atomic<int> host_chunk{8192};     // DIL_HOST_CHUNK: polynomials (KiB) per chunk of the *_host pipelines
    std::atomic<int> host_streams{4};      // DIL_HOST_STREAMS: streams the chunks go round (1 .. 8)
    std::atomic<int> host_duplex{1};       // DIL_HOST_DUPLEX: 1 = page-locked caller buffers: ONE stream carries every upload, one the kernels + downloads (0: chunks round-robin over host_streams)
    std::atomic<int> host_copy_threads{3}; // DIL_HOST_COPY_THREADS: threads (the calling one included) that memcpy between a pageable caller buffer and the page-locked staging slots
    std::atomic<int> multi_group_at_1{0};  // DIL_MULTI_GROUP_AT_1 (tests): 1 | 2 = a one-device dil_*_multi_dev job goes through the grouped collective code
    std::atomic<int> w0w1_plane{1};        // DIL_W0W1_PLANE: 1 = inside the signing loop phase 1 hands w1 to phase 2 in the top byte of the w0 dwords (0: a byte plane of its own)
    std::atomic<int> packed_y{1};          // DIL_PACKED_Y: 1 = the signing loop's large rounds keep y as ExpandMask's raw B-bit stream (0: int32)
};
extern Config cfg;
std::atomic<int>* option_slot(const char* name);     // nullptr: unknown option

// ---- per-stream scratch arenas of the composite calls (scheme.hip) ------------------------------------------
struct Arena {
    hipStream_t stream = nullptr;
    char* base = nullptr;
    size_t size = 0;
    int32_t* pinned = nullptr;      // two host words for small read-backs
    bool in_use = false;
    uint64_t last_use = 0;
};
struct ArenaPool {
    std::mutex mu;
    Arena slots[8];
    uint64_t tick = 0;
    Arena* acquire(hipStream_t s);
    int release(Arena* a, size_t wanted);        // regrows to `wanted` if needed; returns the hipError_t of a failed regrow
    void clear();
};

// helper stream of the composite calls: latency-bound independent parts run beside the caller's stream
struct AuxStream {
    std::mutex mu;
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    bool ensure();
    void destroy();
};

// staging of the *_host transform entry points (capi.hip)
constexpr int HOST_STREAMS = 8;            // upper bound; option host_streams picks how many a call uses
struct HostPipe {
    hipStream_t stream[HOST_STREAMS] = {};
    uint8_t* dev[HOST_STREAMS] = {};       // one staging buffer per stream, allocated when a call first uses that many (ensure_pipe)
    uint8_t* host[HOST_STREAMS] = {};      // page-locked HOST slots of the ring a pageable call goes round (hipHostMalloc; the first HOST_RING only)
    size_t host_bytes[HOST_STREAMS] = {};
    size_t dev_bytes[HOST_STREAMS] = {};   // size of each
    int oversized[HOST_STREAMS] = {};      // consecutive calls that needed less than a quarter of it (or not the buffer at all): given back after 16
    hipEvent_t up_done[HOST_STREAMS] = {}, dn_done[HOST_STREAMS] = {};   // per staging buffer: its upload landed / its download left
    bool ready = false;
};

// the host mailbox of the batch-of-one drop-in calls (kernels.hpp Mailbox; capi.hip mailbox_call)
struct MailboxHost {
    std::mutex mu;                  // one request at a time; a second caller takes the launch path instead of waiting
    dil::Mailbox* host = nullptr;   // pinned, device-mapped
    dil::Mailbox* dev = nullptr;    // the same memory as the device sees it
    hipStream_t stream = nullptr;
    uint32_t seq = 0;
    bool broken = false;            // a request timed out: the mailbox is not used again in this process
    uint64_t launches = 0, calls = 0;
    uint64_t leaked = 0;            // mailboxes left behind by mailbox_destroy because their wave did not retire in time
};

struct Device {
    std::mutex mu;                  // guards (re)initialisation and teardown only
    std::atomic<bool> ready{false};
    int id = -1;
    uint32_t* d_tables = nullptr;   // fwd | inv | inv_pipe
    int num_cus = 256;
    hipMemPool_t pool = nullptr;    // private stream-ordered pool for spills (the application's default pool is left alone)
    std::mutex host_mu;             // serialises the *_host transform entry points of THIS device (they share `scratch`, `hp`)
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    void* stage = nullptr;          // the library's own page-locked staging buffer of the small host-pointer calls (hipHostMalloc)
    size_t stage_bytes = 0;
    HostPipe hp;
    ArenaPool arenas;
    AuxStream aux;
    MailboxHost mbox;
    // launch configuration for a call made now: device constants + the current options
    dil::Tables tables() const;
};

// copies between a caller's HOST buffer and device memory that never hand an unregistered caller pointer to the runtime (capi.hip, the rule
// of the host-pointer entry points): synchronous on the null stream
int host_upload(Device& d, void* dev, const void* host, size_t bytes);
int host_download(Device& d, void* host, const void* dev, size_t bytes);

// The Device record of the calling thread's CURRENT HIP device, initialised on first use.  Every entry point starts
// here, so buffers, stream and tables always belong to one device: the one HIP itself would launch on.
int current(Device** out);
inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace rt
}  // namespace dil

// boilerplate of an entry point: `DIL_ENTER(dv, T)` declares `dil::rt::Device& dv` and `const dil::Tables T`
#define DIL_ENTER(dv, T)                               \
    dil::rt::Device* dv##_ptr = nullptr;               \
    {                                                  \
        const int rc__ = dil::rt::current(&dv##_ptr);  \
        if (rc__) return rc__;                         \
    }                                                  \
    dil::rt::Device& dv = *dv##_ptr;                   \
    const dil::Tables T = dv.tables();                 \
    (void)T
