// multi_gpu.hip -- the C++ multi-GPU host layer (north_star "host code in C++ ... RCCL over xGMI only for the final
// gather").  Independent polynomials / signatures shard embarrassingly: a batch is cut into contiguous slices
// [g*B/G, (g+1)*B/G) (sizes differ by at most one), one host thread per device runs the single-device entry point on its slice
// with that device current -- per-device runtime state (capi_internal.hpp) makes the threads independent -- and there is no
// collective inside the data path.  Two forms:
//   dil_*_multi_host   host buffers: every thread's D2H copy lands in the caller's array (the slabs meet in host memory);
//   dil_*_multi_dev    DEVICE-resident: every device computes its slab in place inside a full-size result array of its own,
//                      then ONE RCCL collective over xGMI completes the arrays -- ncclAllGather when the slices are equal, the
//                      grouped-broadcast form of all-gather-v when they are ragged by one item, or a grouped send / recv to one
//                      root (the shape of SURVEY.md:352; the reference has no counterpart: its only boundary is the 64-bit
//                      stream port of rtl_src/combined_top.v:36-41).
// RCCL is bound at first use with dlopen (the prototypes come from <rccl/rccl.h>): a process that already carries an RCCL --
// PyTorch ships its own copy -- keeps using that one, and libdil256.so loads on hosts without RCCL as long as nobody calls the
// *_multi_dev entry points.  Across processes (one per GPU, torch.distributed) dilithium_amd/sharding.py does the same slicing.
#include "../../include/dil256.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <mutex>
#include <stdio.h>
#include <thread>
#include <vector>

extern "C" void dil_shard_range(size_t n_items, int rank, int world, size_t* lo, size_t* hi)
{
    if (world < 1) world = 1;
    const size_t base = n_items / (size_t)world, rem = n_items % (size_t)world;
    const size_t r = (size_t)rank;
    *lo = r * base + (r < rem ? r : rem);
    *hi = *lo + base + (r < rem ? 1 : 0);
}

namespace {
int device_count(int ndev)
{
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have < 1) return -1;
    return (ndev <= 0 || ndev > have) ? have : ndev;
}

// run fn(g, lo, hi) for every device's slice on its own thread with that device current; first error wins.  A thread that
// cannot be created (std::system_error) ends the fan-out: the ones already running are joined and the error is reported.
template <class F>
int for_each_device(size_t batch, int G, F&& fn)
{
    if (G < 0) return (int)hipErrorNoDevice;
    if (batch == 0) return 0;
    std::vector<int> rc((size_t)G, 0);
    std::vector<std::thread> th;
    int spawn_rc = 0;
    for (int g = 0; g < G; g++) {
        try {
            th.emplace_back([&, g] {
                size_t lo, hi;
                dil_shard_range(batch, g, G, &lo, &hi);
                if (lo == hi) return;
                const hipError_t e = hipSetDevice(g);
                rc[(size_t)g] = e != hipSuccess ? (int)e : fn(g, lo, hi);
            });
        } catch (...) {
            spawn_rc = (int)hipErrorOutOfMemory;
            break;
        }
    }
    for (std::thread& t : th) t.join();
    if (spawn_rc) return spawn_rc;
    for (int r : rc)
        if (r) return r;
    return 0;
}

// ---- RCCL, bound at first use ---------------------------------------------------------------------------------------
// The handful of NCCL-API declarations this file needs, stated here instead of #include <rccl/rccl.h>: the library then builds
// on hosts without the RCCL headers and cannot pick up prototypes that differ from the library it finds at run time -- these are
// the stable NCCL 2.x C ABI (opaque communicator pointer, int-sized enums), and ncclGetVersion() is checked against it below
// (round-3 advisor finding).
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;                    // ncclSuccess = 0
typedef int ncclDataType_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclDataType_t ncclUint8 = 1;      // nccl.h: ncclInt8 = 0, ncclUint8 = 1
struct Rccl {
    void* so = nullptr;
    int version = 0;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    const char* why = "";
    bool load()
    {
        if (so) return true;
        void* h = nullptr;
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names)                      // an RCCL this process already carries (PyTorch's) wins
            if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
        const char* paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
        for (int i = 0; !h && i < 3; i++) h = dlopen(paths[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            why = "librccl not found";
            return false;
        }
#define DIL_SYM(f)                                                      \
    f = reinterpret_cast<decltype(f)>(dlsym(h, "nccl" #f));             \
    if (!f) {                                                           \
        why = "librccl lacks nccl" #f;                                  \
        dlclose(h);                                                     \
        return false;                                                   \
    }
        DIL_SYM(GetVersion) DIL_SYM(CommInitAll) DIL_SYM(CommDestroy) DIL_SYM(CommAbort) DIL_SYM(AllGather) DIL_SYM(Broadcast) DIL_SYM(Send)
        DIL_SYM(Recv) DIL_SYM(GroupStart) DIL_SYM(GroupEnd) DIL_SYM(GetErrorString)
#undef DIL_SYM
        // NCCL_VERSION_CODE = major * 10000 + minor * 100 + patch from 2.9 on (major * 1000 + ... before): the declarations above are
        // the 2.x ABI with ncclSend / ncclRecv (2.7+)
        if (GetVersion(&version) != ncclSuccess || version < 2700 || (version >= 10000 && version / 10000 != 2)) {
            why = "unsupported RCCL version (need the NCCL 2.x API, 2.7 or later)";
            dlclose(h);
            return false;
        }
        so = h;
        return true;
    }
};

struct Multi {
    std::mutex call_mu;                 // one *_multi_dev call (compute on the devices' streams + its collectives) at a time
    std::mutex mu;                      // the state below
    Rccl rccl;
    int G = 0;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    char last_error[256] = "";
};
Multi g_multi;

int rccl_fail(ncclResult_t r, const char* what)
{
    snprintf(g_multi.last_error, sizeof(g_multi.last_error), "%s: %s", what, g_multi.rccl.GetErrorString ? g_multi.rccl.GetErrorString(r) : "?");
    fprintf(stderr, "libdil256: RCCL %s\n", g_multi.last_error);
    return DIL_ERR_RCCL;
}
#define DIL_NCCL(call, what)                          \
    do {                                              \
        const ncclResult_t r__ = (call);              \
        if (r__ != ncclSuccess) return rccl_fail(r__, what); \
    } while (0)

void multi_teardown_locked()
{
    Multi& m = g_multi;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (int g = 0; g < m.G; g++) {
        if (hipSetDevice(g) != hipSuccess) continue;
        if (m.stream[(size_t)g]) (void)hipStreamDestroy(m.stream[(size_t)g]);
        if (m.comm[(size_t)g] && m.rccl.CommDestroy) (void)m.rccl.CommDestroy(m.comm[(size_t)g]);
    }
    m.comm.clear();
    m.stream.clear();
    m.G = 0;
    if (have_cur) (void)hipSetDevice(cur);
}

// communicators + one stream per device over devices 0 .. G-1; rebuilt when G changes
int multi_ensure(int ndev, int* G_out)
{
    Multi& m = g_multi;
    const int G = device_count(ndev);
    if (G < 0) return (int)hipErrorNoDevice;
    std::lock_guard<std::mutex> lk(m.mu);
    *G_out = G;
    if (m.G == G) return 0;
    if (m.G) multi_teardown_locked();
    if (!m.rccl.load()) {
        const char* dl = dlerror();
        snprintf(m.last_error, sizeof(m.last_error), "%s%s%s", m.rccl.why, dl ? ": " : "", dl ? dl : "");
        return DIL_ERR_RCCL;
    }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    std::vector<int> devs((size_t)G);
    for (int g = 0; g < G; g++) devs[(size_t)g] = g;
    m.comm.assign((size_t)G, nullptr);
    m.stream.assign((size_t)G, nullptr);
    DIL_NCCL(m.rccl.CommInitAll(m.comm.data(), G, devs.data()), "ncclCommInitAll");
    m.G = G;
    for (int g = 0; g < G; g++) {
        int e = (int)hipSetDevice(g);
        if (!e) e = (int)hipStreamCreateWithFlags(&m.stream[(size_t)g], hipStreamNonBlocking);
        if (!e) e = dil_init(-1);                                      // this device's runtime state (tables, pool) up front
        if (e) {
            multi_teardown_locked();
            if (have_cur) (void)hipSetDevice(cur);
            return e;
        }
    }
    if (have_cur) (void)hipSetDevice(cur);
    return 0;
}

// The schedule of the final gather, device by device in issue order (one group): in-place all-gather of equal slabs; ragged slabs (by one
// item) as all-gather-v = one broadcast per slab; gather to one root as its receives of every other slab and every other device's send of
// its own.  Pure: no device, no RCCL.
void gather_plan(size_t batch, size_t item_bytes, int root, int G, bool force_ragged, std::vector<dil_gather_op>& ops)
{
    ops.clear();
    if (G < 1 || batch == 0 || item_bytes == 0) return;
    const bool equal = batch % (size_t)G == 0 && !force_ragged;
    for (int g = 0; g < G; g++) {
        if (root < 0 && equal) {                     // the send slab sits at its own offset of the receive array
            const size_t cnt = batch / (size_t)G * item_bytes;
            ops.push_back({DIL_GATHER_ALLGATHER, g, -1, (size_t)g * cnt, cnt});
        } else if (root < 0) {
            for (int r = 0; r < G; r++) {
                size_t lo, hi;
                dil_shard_range(batch, r, G, &lo, &hi);
                if (hi > lo) ops.push_back({DIL_GATHER_BROADCAST, g, r, lo * item_bytes, (hi - lo) * item_bytes});
            }
        } else if (g == root) {                      // the root receives every other slab ...
            for (int r = 0; r < G; r++) {
                size_t lo, hi;
                dil_shard_range(batch, r, G, &lo, &hi);
                if (r != root && hi > lo) ops.push_back({DIL_GATHER_RECV, g, r, lo * item_bytes, (hi - lo) * item_bytes});
            }
        } else {                                     // ... and every other device sends its own
            size_t lo, hi;
            dil_shard_range(batch, g, G, &lo, &hi);
            if (hi > lo) ops.push_back({DIL_GATHER_SEND, g, root, lo * item_bytes, (hi - lo) * item_bytes});
        }
    }
}

// THE collective of the design: every device g holds items [lo_g, hi_g) of `bufs[g]` ([batch][item_bytes], device memory of
// device g); afterwards every bufs[g] (root < 0) or bufs[root] alone holds all items.  Enqueued on the devices' streams, which
// are then drained.
int gather_slabs(void* const* bufs, size_t item_bytes, size_t batch, int root, int G)
{
    Multi& m = g_multi;
    std::lock_guard<std::mutex> lk(m.mu);
    if (m.G != G) return (int)hipErrorInvalidValue;
    if (root >= G) return (int)hipErrorInvalidValue;
    // Option multi_group_at_1 (tests only): send a ONE-device job through the grouped code below instead of the trivial collective --
    // 1: GroupStart, the in-place ncclAllGather, GroupEnd, drain; 2: the ragged form, one ncclBroadcast per slab inside the group.  On a
    // one-GPU box this is the only way any of the group code runs before a multi-GPU node sees it.  (ncclSend / ncclRecv of the
    // gather-to-a-root form need two ranks: they execute for the first time there.)
    int force = 0;
    if (G == 1 && dil_get_option("multi_group_at_1", &force) != 0) force = 0;
    if ((G > 1 || force) && batch > 0 && item_bytes > 0) {
        std::vector<dil_gather_op> plan;
        gather_plan(batch, item_bytes, root, G, force == 2, plan);
        DIL_NCCL(m.rccl.GroupStart(), "ncclGroupStart");
        ncclResult_t bad = ncclSuccess;                  // first failure inside the group; the group is closed either way
        const char* bad_what = "";
#define DIL_IN_GROUP(call, what)                                  \
    do {                                                          \
        const ncclResult_t r__ = bad == ncclSuccess ? (call) : bad; \
        if (r__ != ncclSuccess && bad == ncclSuccess) {           \
            bad = r__;                                            \
            bad_what = what;                                      \
        }                                                         \
    } while (0)
        for (const dil_gather_op& op : plan) {           // the schedule is pure host logic (gather_plan, exported as dil_multi_gather_plan and tested without a GPU)
            const size_t g = (size_t)op.rank;
            char* mine = static_cast<char*>(bufs[g]);
            switch (op.kind) {
            case DIL_GATHER_ALLGATHER:
                DIL_IN_GROUP(m.rccl.AllGather(mine + op.offset, mine, op.bytes, ncclUint8, m.comm[g], m.stream[g]), "ncclAllGather");
                break;
            case DIL_GATHER_BROADCAST:
                DIL_IN_GROUP(m.rccl.Broadcast(mine + op.offset, mine + op.offset, op.bytes, ncclUint8, op.peer, m.comm[g], m.stream[g]), "ncclBroadcast");
                break;
            case DIL_GATHER_RECV:
                DIL_IN_GROUP(m.rccl.Recv(mine + op.offset, op.bytes, ncclUint8, op.peer, m.comm[g], m.stream[g]), "ncclRecv");
                break;
            default:
                DIL_IN_GROUP(m.rccl.Send(mine + op.offset, op.bytes, ncclUint8, op.peer, m.comm[g], m.stream[g]), "ncclSend");
                break;
            }
        }
#undef DIL_IN_GROUP
        const ncclResult_t ge = m.rccl.GroupEnd();
        if (bad != ncclSuccess || ge != ncclSuccess) {
            // A group that was closed with only part of its operations enqueued can leave the ranks that did enqueue waiting for the
            // ones that did not: abort every communicator (ncclCommAbort also releases work already on the streams), drain the
            // streams, and drop the state so that the next call builds fresh communicators (round-3 advisor finding).
            int cur = 0;
            const bool have_cur = hipGetDevice(&cur) == hipSuccess;
            for (int g = 0; g < G; g++) {
                if (m.comm[(size_t)g]) (void)m.rccl.CommAbort(m.comm[(size_t)g]);
                m.comm[(size_t)g] = nullptr;
            }
            for (int g = 0; g < G; g++)
                if (hipSetDevice(g) == hipSuccess) (void)hipStreamSynchronize(m.stream[(size_t)g]);
            if (have_cur) (void)hipSetDevice(cur);
            multi_teardown_locked();
            return bad != ncclSuccess ? rccl_fail(bad, bad_what) : rccl_fail(ge, "ncclGroupEnd");
        }
    } else if (G == 1 && batch > 0 && item_bytes > 0) {
        // one device: the slab IS the array; still one (trivial) RCCL collective so that the path is the one a node runs
        DIL_NCCL(m.rccl.AllGather(bufs[0], bufs[0], batch * item_bytes, ncclUint8, m.comm[0], m.stream[0]), "ncclAllGather");
    }
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    int rc = 0;
    for (int g = 0; g < G; g++) {
        hipError_t e = hipSetDevice(g);
        if (e == hipSuccess) e = hipStreamSynchronize(m.stream[(size_t)g]);
        if (e != hipSuccess && !rc) rc = (int)e;
    }
    if (have_cur) (void)hipSetDevice(cur);
    return rc;
}

hipStream_t dev_stream(int g) { return g_multi.stream[(size_t)g]; }
}  // namespace

extern "C" {

// What the collective layer is bound to: NCCL_VERSION_CODE of the RCCL in use (major * 10000 + minor * 100 + patch), the path of the
// library (as the dynamic linker reports it), communicators alive.  0 on success; RCCL is bound on first use (DIL_ERR_RCCL if absent).
int dil_multi_info(int* rccl_version, char* path, size_t path_len, int* ndev)
{
    std::lock_guard<std::mutex> lk(g_multi.mu);
    if (!g_multi.rccl.load()) {
        const char* dl = dlerror();
        snprintf(g_multi.last_error, sizeof(g_multi.last_error), "%s%s%s", g_multi.rccl.why, dl ? ": " : "", dl ? dl : "");
        return DIL_ERR_RCCL;
    }
    if (rccl_version) *rccl_version = g_multi.rccl.version;
    if (ndev) *ndev = g_multi.G;
    if (path && path_len) {
        path[0] = 0;
        Dl_info info;
        if (dladdr(reinterpret_cast<void*>(g_multi.rccl.AllGather), &info) && info.dli_fname) snprintf(path, path_len, "%s", info.dli_fname);
    }
    return 0;
}

int dil_multi_gather_plan(size_t batch, size_t item_bytes, int gather_root, int ndev, int force_ragged, dil_gather_op* ops, size_t max_ops, size_t* n_ops)
{
    if (ndev < 1 || gather_root >= ndev || !n_ops) return (int)hipErrorInvalidValue;
    std::vector<dil_gather_op> plan;
    gather_plan(batch, item_bytes, gather_root, ndev, force_ragged != 0, plan);
    *n_ops = plan.size();
    for (size_t i = 0; ops && i < plan.size() && i < max_ops; i++) ops[i] = plan[i];
    return 0;
}

const char* dil_multi_last_error(void)
{
    static thread_local char copy[256];
    std::lock_guard<std::mutex> lk(g_multi.mu);
    snprintf(copy, sizeof(copy), "%s", g_multi.last_error);
    return copy;
}

int dil_multi_init(int ndev)
{
    std::lock_guard<std::mutex> call(g_multi.call_mu);
    int G;
    return multi_ensure(ndev, &G);
}

int dil_multi_shutdown(void)
{
    std::lock_guard<std::mutex> call(g_multi.call_mu);
    std::lock_guard<std::mutex> lk(g_multi.mu);
    multi_teardown_locked();
    return 0;
}

int dil_gather_slabs_multi_dev(void* const* bufs, size_t item_bytes, size_t batch, int gather_root, int ndev)
{
    std::lock_guard<std::mutex> call(g_multi.call_mu);
    int G, rc;
    if (!bufs) return (int)hipErrorInvalidValue;
    if ((rc = multi_ensure(ndev, &G))) return rc;
    return gather_slabs(bufs, item_bytes, batch, gather_root, G);
}

int dil_ntt_multi_dev(int32_t* const* polys, size_t batch, int inverse, int gather_root, int ndev)
{
    std::lock_guard<std::mutex> call(g_multi.call_mu);
    int G, rc;
    if (!polys) return (int)hipErrorInvalidValue;
    if ((rc = multi_ensure(ndev, &G))) return rc;
    rc = for_each_device(batch, G, [&](int g, size_t lo, size_t hi) {
        int32_t* slab = polys[g] + lo * 256;
        return inverse ? dil_invntt_dev(slab, hi - lo, dev_stream(g)) : dil_ntt_dev(slab, hi - lo, dev_stream(g));
    });
    if (rc) return rc;
    return gather_slabs(reinterpret_cast<void* const*>(polys), 1024, batch, gather_root, G);
}

int dil_sign_multi_dev(uint8_t* const* sig, int32_t* const* attempts, const uint8_t* const* sk, const uint8_t* const* mu, int level,
                       size_t batch, int shared_sk, int max_attempts, int gather_root, int ndev)
{
    std::lock_guard<std::mutex> call(g_multi.call_mu);
    const size_t sgb = dil_sig_bytes(level);
    int G, rc;
    if (!sgb || !sig || !sk || !mu) return (int)hipErrorInvalidValue;
    if ((rc = multi_ensure(ndev, &G))) return rc;
    bool unfinished = false;
    std::mutex um;
    rc = for_each_device(batch, G, [&](int g, size_t lo, size_t hi) {
        const int r = dil_sign_dev(sig[g] + lo * sgb, attempts ? attempts[g] + lo : nullptr, sk[g], mu[g], level, hi - lo, shared_sk, max_attempts,
                                   dev_stream(g));
        if (r == DIL_ERR_UNFINISHED) {
            std::lock_guard<std::mutex> lk(um);
            unfinished = true;
            return 0;
        }
        return r;
    });
    if (rc) return rc;
    if ((rc = gather_slabs(reinterpret_cast<void* const*>(sig), sgb, batch, gather_root, G))) return rc;
    if (attempts && (rc = gather_slabs(reinterpret_cast<void* const*>(attempts), 4, batch, gather_root, G))) return rc;
    return unfinished ? DIL_ERR_UNFINISHED : 0;
}

int dil_verify_sig_multi_dev(int32_t* const* verdict, const uint8_t* const* pk, const uint8_t* const* sig, const uint8_t* const* mu, int level,
                             size_t batch, int shared_pk, int gather_root, int ndev)
{
    std::lock_guard<std::mutex> call(g_multi.call_mu);
    int G, rc;
    if (!dil_pk_bytes(level) || !verdict || !pk || !sig || !mu) return (int)hipErrorInvalidValue;
    if ((rc = multi_ensure(ndev, &G))) return rc;
    rc = for_each_device(batch, G, [&](int g, size_t lo, size_t hi) {
        return dil_verify_sig_dev(verdict[g] + lo, pk[g], sig[g], mu[g], level, hi - lo, shared_pk, dev_stream(g));
    });
    if (rc) return rc;
    return gather_slabs(reinterpret_cast<void* const*>(verdict), 4, batch, gather_root, G);
}

// BASELINE configs[4] as north_star states it: the level-5 sign inner loop (phase 1 + phase 2) on every device's slice of the
// attempts, then the gather of the (z, h, flag) slabs.  A / s1hat / s2hat / t0hat: this device's copy of the (shared) key
// material or its slice of per-attempt keys; y, c: this device's slice; z / h / flags: full-size arrays, slab written in place.
int dil_sign_phases_multi_dev(int32_t* const* z, uint8_t* const* h, int32_t* const* flags, const int32_t* const* A, const int32_t* const* y,
                              const int32_t* const* c, const int32_t* const* s1hat, const int32_t* const* s2hat, const int32_t* const* t0hat,
                              uint8_t* const* w1_scratch, int32_t* const* w0_scratch, int level, size_t batch, int shared_key, int gather_root,
                              int ndev)
{
    std::lock_guard<std::mutex> call(g_multi.call_mu);
    int G, rc;
    const int K = level == 2 ? 4 : level == 3 ? 6 : level == 5 ? 8 : 0, L = level == 2 ? 4 : level == 3 ? 5 : 7;
    if (!K || !z || !h || !flags || !A || !y || !c || !s1hat || !s2hat || !t0hat || !w1_scratch || !w0_scratch) return (int)hipErrorInvalidValue;
    if ((rc = multi_ensure(ndev, &G))) return rc;
    rc = for_each_device(batch, G, [&](int g, size_t lo, size_t hi) {
        int r = dil_sign_phase1_dev(w1_scratch[g], w0_scratch[g], A[g], y[g], level, hi - lo, shared_key, dev_stream(g));
        if (r) return r;
        return dil_sign_phase2_dev(z[g] + lo * (size_t)L * 256, h[g] + lo * (size_t)K * 256, flags[g] + lo, c[g], y[g], w0_scratch[g], w1_scratch[g],
                                   s1hat[g], s2hat[g], t0hat[g], level, hi - lo, shared_key, dev_stream(g));
    });
    if (rc) return rc;
    if ((rc = gather_slabs(reinterpret_cast<void* const*>(z), (size_t)L * 1024, batch, gather_root, G))) return rc;
    if ((rc = gather_slabs(reinterpret_cast<void* const*>(h), (size_t)K * 256, batch, gather_root, G))) return rc;
    return gather_slabs(reinterpret_cast<void* const*>(flags), 4, batch, gather_root, G);
}

// ---- host-buffer forms: the slabs meet in host memory, no collective -----------------------------------------------------
int dil_ntt_multi_host(int32_t* polys, size_t batch, int inverse, int ndev)
{
    return for_each_device(batch, device_count(ndev), [&](int, size_t lo, size_t hi) {
        return inverse ? dil_invntt_host(polys + lo * 256, hi - lo) : dil_ntt_host(polys + lo * 256, hi - lo);
    });
}

int dil_keygen_multi_host(uint8_t* pk, uint8_t* sk, const uint8_t* seed, int level, size_t batch, int ndev)
{
    const size_t pkb = dil_pk_bytes(level), skb = dil_sk_bytes(level);
    if (!pkb) return (int)hipErrorInvalidValue;
    return for_each_device(batch, device_count(ndev), [&](int, size_t lo, size_t hi) {
        return dil_keygen_host(pk + lo * pkb, sk + lo * skb, seed + lo * 32, level, hi - lo);
    });
}

int dil_sign_multi_host(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* mu, int level, size_t batch, int shared_sk,
                        int max_attempts, int ndev)
{
    const size_t skb = dil_sk_bytes(level), sgb = dil_sig_bytes(level);
    if (!skb) return (int)hipErrorInvalidValue;
    return for_each_device(batch, device_count(ndev), [&](int, size_t lo, size_t hi) {
        return dil_sign_host(sig + lo * sgb, attempts ? attempts + lo : nullptr, shared_sk ? sk : sk + lo * skb, mu + lo * 64, level,
                             hi - lo, shared_sk, max_attempts);
    });
}

int dil_verify_sig_multi_host(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level, size_t batch,
                              int shared_pk, int ndev)
{
    const size_t pkb = dil_pk_bytes(level), sgb = dil_sig_bytes(level);
    if (!pkb) return (int)hipErrorInvalidValue;
    return for_each_device(batch, device_count(ndev), [&](int, size_t lo, size_t hi) {
        return dil_verify_sig_host(verdict + lo, shared_pk ? pk : pk + lo * pkb, sig + lo * sgb, mu + lo * 64, level, hi - lo, shared_pk);
    });
}

}  // extern "C"
