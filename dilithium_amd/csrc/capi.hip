// capi.hip -- the extern "C" boundary of libdil256.so (declared in include/dil256.h) and the
// host runtime behind it: twiddle-table construction, device selection, scratch management
// for the host-pointer entry points.  Host language is C++ because the reference's
// dilithium-256/ is C++; nothing here is a CPU fallback -- every arithmetic entry
// point launches a HIP kernel and returns the hipError_t if that is not possible.
#include "../../include/dil256.h"
#include "kernels.hpp"

#include <algorithm>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int64_t Q = DIL_Q;

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

struct State {
    std::mutex mu;
    bool ready = false;
    int device = -1;
    uint32_t* d_tables = nullptr;   // fwd | inv | inv_pipe
    dil::Tables t;
    void* scratch = nullptr;        // for *_host entry points
    size_t scratch_bytes = 0;
    int sign_cap = 0;               // DIL_SIGN_CAP: entries in flight per signing round (0 = default)
    int aux_overlap = 1;            // DIL_AUX_OVERLAP: 0 = composite calls never use the helper stream
    int sign_streams = 1;           // DIL_SIGN_STREAMS: 2 = split each signing round over the caller stream and a helper (measured: no gain)
};
State g;

#define DIL_TRY(expr)                          \
    do {                                       \
        hipError_t e__ = (expr);               \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

int ensure_init()
{
    if (g.ready) return 0;
    return dil_init(-1);
}

int ensure_scratch(size_t bytes)
{
    if (bytes <= g.scratch_bytes) return 0;
    if (g.scratch) {
        DIL_TRY(hipFree(g.scratch));
        g.scratch = nullptr;
        g.scratch_bytes = 0;
    }
    DIL_TRY(hipMalloc(&g.scratch, bytes));
    g.scratch_bytes = bytes;
    return 0;
}

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

// host wrapper: the reference's callers hold HOST buffers.  Small batches: copy in, run, copy
// out on the default stream.  Large batches: chunks of HOST_CHUNK polynomials round-robin over
// HOST_STREAMS streams, each chunk H2D -> kernel -> D2H on its own stream, so the PCIe transfers of
// one chunk overlap the kernel and the opposite-direction transfer of its neighbours.  With
// DIL_HOST_PIN=1 the caller's buffer is page-locked (hipHostRegister) for the duration of the call
// so that the copies are true asynchronous DMA.
constexpr size_t HOST_CHUNK = 16384;     // polynomials per chunk (16 MiB)
constexpr int HOST_STREAMS = 3;

struct HostPipe {
    hipStream_t stream[HOST_STREAMS] = {nullptr, nullptr, nullptr};
    int32_t* dev[HOST_STREAMS] = {nullptr, nullptr, nullptr};
    bool ready = false;
    int pin = -1;
};
HostPipe hp;

int ensure_pipe()
{
    if (hp.ready) return 0;
    for (int i = 0; i < HOST_STREAMS; i++) {
        DIL_TRY(hipStreamCreateWithFlags(&hp.stream[i], hipStreamNonBlocking));
        DIL_TRY(hipMalloc(reinterpret_cast<void**>(&hp.dev[i]), HOST_CHUNK * 1024));
    }
    const char* e = getenv("DIL_HOST_PIN");
    hp.pin = (e && atoi(e) != 0) ? 1 : 0;
    hp.ready = true;
    return 0;
}

template <class F>
int host_inplace(int32_t* h, size_t batch, F&& fn)   // fn(device_ptr, n_polys, stream) -> int
{
    if (batch == 0) return 0;
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g.mu);
    if (batch <= HOST_CHUNK) {
        const size_t bytes = batch * 1024;
        rc = ensure_scratch(bytes);
        if (rc) return rc;
        DIL_TRY(hipMemcpy(g.scratch, h, bytes, hipMemcpyHostToDevice));
        rc = fn(static_cast<int32_t*>(g.scratch), batch, (hipStream_t)0);
        if (rc) return rc;
        DIL_TRY(hipDeviceSynchronize());
        DIL_TRY(hipMemcpy(h, g.scratch, bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    rc = ensure_pipe();
    if (rc) return rc;
    bool pinned = false;
    if (hp.pin) pinned = hipHostRegister(h, batch * 1024, hipHostRegisterDefault) == hipSuccess;
    int err = 0;
    size_t c = 0;
    for (size_t off = 0; off < batch && !err; off += HOST_CHUNK, c++) {
        const int s = (int)(c % HOST_STREAMS);
        const size_t n = batch - off < HOST_CHUNK ? batch - off : HOST_CHUNK;
        int32_t* hc = h + off * 256;
        err = (int)hipMemcpyAsync(hp.dev[s], hc, n * 1024, hipMemcpyHostToDevice, hp.stream[s]);
        if (!err) err = fn(hp.dev[s], n, hp.stream[s]);
        if (!err) err = (int)hipMemcpyAsync(hc, hp.dev[s], n * 1024, hipMemcpyDeviceToHost, hp.stream[s]);
    }
    for (int i = 0; i < HOST_STREAMS; i++) {
        const hipError_t e = hipStreamSynchronize(hp.stream[i]);
        if (!err && e != hipSuccess) err = (int)e;
    }
    if (pinned) (void)hipHostUnregister(h);
    return err;
}

}  // namespace

extern "C" {

void dil_host_twiddle_tables(uint32_t* fwd, uint32_t* inv, uint32_t* inv_pipe) { build_tables(fwd, inv, inv_pipe); }

void dil_host_zetas(int32_t* zetas)
{
    uint32_t z[256];
    canonical_zetas(z);
    for (int k = 0; k < 256; k++) zetas[k] = (int32_t)(z[k] > (uint32_t)(Q - 1) / 2 ? (int64_t)z[k] - Q : z[k]);
}

int dil_device_count(int* count) { return (int)hipGetDeviceCount(count); }

int dil_num_cus(void) { return g.ready ? g.t.num_cus : -1; }

const char* dil_error_string(int code)
{
    if (code == DIL_ERR_UNFINISHED) return "signing did not finish within max_attempts";
    return hipGetErrorString((hipError_t)code);
}

int dil_init(int device)
{
    std::lock_guard<std::mutex> lk(g.mu);
    int cur = 0;
    if (device < 0) {
        DIL_TRY(hipGetDevice(&cur));
        device = cur;
    }
    if (g.ready && g.device == device) return 0;
    DIL_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    DIL_TRY(hipGetDeviceProperties(&prop, device));
    if (g.d_tables) {
        (void)hipFree(g.d_tables);
        g.d_tables = nullptr;
    }
    if (g.scratch) {
        (void)hipFree(g.scratch);
        g.scratch = nullptr;
        g.scratch_bytes = 0;
    }
    static uint32_t h_tab[3 * 2048];
    build_tables(h_tab, h_tab + 2048, h_tab + 4096);
    DIL_TRY(hipMalloc(reinterpret_cast<void**>(&g.d_tables), sizeof(h_tab)));
    DIL_TRY(hipMemcpy(g.d_tables, h_tab, sizeof(h_tab), hipMemcpyHostToDevice));
    g.t.fwd = g.d_tables;
    g.t.inv = g.d_tables + 2048;
    g.t.inv_pipe = g.d_tables + 4096;
    g.t.num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (const char* e = getenv("DIL_NTT_BPC")) g.t.ntt_blocks_per_cu = atoi(e) > 0 ? atoi(e) : g.t.ntt_blocks_per_cu;
    if (const char* e = getenv("DIL_WPI_BPC")) g.t.wpi_blocks_per_cu = atoi(e) > 0 ? atoi(e) : g.t.wpi_blocks_per_cu;
    if (const char* e = getenv("DIL_FUSED_MODE")) g.t.fused_mode = atoi(e);
    if (const char* e = getenv("DIL_SIGN_CAP")) g.sign_cap = atoi(e);
    if (const char* e = getenv("DIL_SIGN_STREAMS")) g.sign_streams = atoi(e);
    if (const char* e = getenv("DIL_AUX_OVERLAP")) g.aux_overlap = atoi(e);
    if (const char* e = getenv("DIL_FUSED_WGPC")) g.t.fused_wgs_per_cu = atoi(e) > 0 ? atoi(e) : g.t.fused_wgs_per_cu;
    {   // composite entry points take their temporaries from the stream-ordered pool: keep what it has
        // grown to instead of handing it back to the driver at every synchronisation
        hipMemPool_t pool;
        if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess) {
            uint64_t keep = getenv("DIL_POOL_KEEP") ? strtoull(getenv("DIL_POOL_KEEP"), nullptr, 10) : (uint64_t)8 << 30;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
    }
    g.device = device;
    g.ready = true;
    return 0;
}

int dil_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.ready) return 0;
    if (g.d_tables) (void)hipFree(g.d_tables);
    if (g.scratch) (void)hipFree(g.scratch);
    if (hp.ready) {
        for (int i = 0; i < HOST_STREAMS; i++) {
            (void)hipFree(hp.dev[i]);
            (void)hipStreamDestroy(hp.stream[i]);
        }
        hp = HostPipe{};
    }
    g.d_tables = nullptr;
    g.scratch = nullptr;
    g.scratch_bytes = 0;
    g.ready = false;
    return 0;
}

// ---- transforms ---------------------------------------------------------------------------
int dil_ntt_dev(int32_t* polys, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_ntt(false, dil::LAYOUT_POLY, 0, polys, batch, g.t, S(stream));
}
int dil_invntt_dev(int32_t* polys, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_ntt(true, dil::LAYOUT_POLY, 0, polys, batch, g.t, S(stream));
}
int dil_ntt_host(int32_t* polys, size_t batch)
{
    return host_inplace(polys, batch, [&](int32_t* d, size_t n, hipStream_t st) { return (int)dil::launch_ntt(false, dil::LAYOUT_POLY, 0, d, n, g.t, st); });
}
int dil_invntt_host(int32_t* polys, size_t batch)
{
    return host_inplace(polys, batch, [&](int32_t* d, size_t n, hipStream_t st) { return (int)dil::launch_ntt(true, dil::LAYOUT_POLY, 0, d, n, g.t, st); });
}

// ---- element-wise ---------------------------------------------------------------------------
int dil_pointwise_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_pointwise(dil::OP_MUL, c, a, b, nullptr, batch, g.t, S(stream));
}
int dil_pointwise_acc_dev(int32_t* c, const int32_t* acc, const int32_t* a, const int32_t* b, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_pointwise(dil::OP_MAC, c, a, b, acc, batch, g.t, S(stream));
}
int dil_poly_add_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_pointwise(dil::OP_ADD, c, a, b, nullptr, batch, g.t, S(stream));
}
int dil_poly_sub_dev(int32_t* c, const int32_t* a, const int32_t* b, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_pointwise(dil::OP_SUB, c, a, b, nullptr, batch, g.t, S(stream));
}
int dil_pointwise_host(int32_t* c, const int32_t* a, const int32_t* b, size_t batch)
{
    if (batch == 0) return 0;
    int rc = ensure_init();
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(g.mu);
    const size_t bytes = batch * 1024;
    rc = ensure_scratch(2 * bytes);
    if (rc) return rc;
    int32_t* da = static_cast<int32_t*>(g.scratch);
    int32_t* db = da + batch * 256;
    DIL_TRY(hipMemcpy(da, a, bytes, hipMemcpyHostToDevice));
    DIL_TRY(hipMemcpy(db, b, bytes, hipMemcpyHostToDevice));
    DIL_TRY(dil::launch_pointwise(dil::OP_MUL, da, da, db, nullptr, batch, g.t, 0));
    DIL_TRY(hipDeviceSynchronize());
    DIL_TRY(hipMemcpy(c, da, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// ---- bram (hardware-model API) -----------------------------------------------------------------
static int check_mapping(int m) { return (m < 0 || m > 2) ? (int)hipErrorInvalidValue : 0; }

int dil_bram_fwdntt_dev(int32_t* ram, size_t batch, int mapping, void* stream)
{
    int rc = ensure_init();
    if (rc || (rc = check_mapping(mapping))) return rc;
    return (int)dil::launch_ntt(false, dil::LAYOUT_BRAM, mapping, ram, batch, g.t, S(stream));
}
int dil_bram_invntt_dev(int32_t* ram, size_t batch, int mapping, void* stream)
{
    int rc = ensure_init();
    if (rc || (rc = check_mapping(mapping))) return rc;
    return (int)dil::launch_ntt(true, dil::LAYOUT_BRAM, mapping, ram, batch, g.t, S(stream));
}
int dil_bram_mul_dev(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping, void* stream)
{
    int rc = ensure_init();
    if (rc || (rc = check_mapping(mapping))) return rc;
    return (int)dil::launch_bram_mul(ram, mul_ram, batch, mapping, g.t, S(stream));
}
int dil_bram_fwdntt_host(int32_t* ram, size_t batch, int mapping)
{
    int rc = check_mapping(mapping);
    if (rc) return rc;
    return host_inplace(ram, batch, [&](int32_t* d, size_t n, hipStream_t st) { return (int)dil::launch_ntt(false, dil::LAYOUT_BRAM, mapping, d, n, g.t, st); });
}
int dil_bram_invntt_host(int32_t* ram, size_t batch, int mapping)
{
    int rc = check_mapping(mapping);
    if (rc) return rc;
    return host_inplace(ram, batch, [&](int32_t* d, size_t n, hipStream_t st) { return (int)dil::launch_ntt(true, dil::LAYOUT_BRAM, mapping, d, n, g.t, st); });
}
int dil_bram_mul_host(int32_t* ram, const int32_t* mul_ram, size_t batch, int mapping)
{
    if (batch == 0) return 0;
    int rc = check_mapping(mapping);
    if (rc || (rc = ensure_init())) return rc;
    std::lock_guard<std::mutex> lk(g.mu);
    const size_t bytes = batch * 1024;
    rc = ensure_scratch(2 * bytes);
    if (rc) return rc;
    int32_t* da = static_cast<int32_t*>(g.scratch);
    int32_t* db = da + batch * 256;
    DIL_TRY(hipMemcpy(da, ram, bytes, hipMemcpyHostToDevice));
    DIL_TRY(hipMemcpy(db, mul_ram, bytes, hipMemcpyHostToDevice));
    DIL_TRY(dil::launch_bram_mul(da, db, batch, mapping, g.t, 0));
    DIL_TRY(hipDeviceSynchronize());
    DIL_TRY(hipMemcpy(ram, da, bytes, hipMemcpyDeviceToHost));
    return 0;
}

// ---- fused pipelines ---------------------------------------------------------------------------
int dil_matvec_dev(int32_t* w, const int32_t* A, const int32_t* y, int level, size_t batch, int shared_A, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_matvec(level, dil::OUT_W, w, nullptr, nullptr, A, y, batch, shared_A, g.t, S(stream));
}
int dil_verify_core_dev(uint8_t* w1, const int32_t* A, const int32_t* z, const int32_t* c, const int32_t* t1,
                        const uint8_t* h, int level, size_t batch, int shared_pk, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_verify(level, w1, A, z, c, t1, h, batch, shared_pk, g.t, S(stream));
}
int dil_sign_phase1_dev(uint8_t* w1, int32_t* w0, const int32_t* A, const int32_t* y, int level, size_t batch,
                        int shared_key, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_matvec(level, dil::OUT_W1W0, nullptr, w1, w0, A, y, batch, shared_key, g.t, S(stream));
}
int dil_sign_phase2_dev(int32_t* z, uint8_t* h, int32_t* flags, const int32_t* c, const int32_t* y, const int32_t* w0,
                        const uint8_t* w1, const int32_t* s1hat, const int32_t* s2hat, const int32_t* t0hat, int level,
                        size_t batch, int shared_key, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_sign2(level, z, h, flags, c, y, w0, w1, s1hat, s2hat, t0hat, batch, shared_key, g.t, S(stream));
}

// ---- row N1: samplers ---------------------------------------------------------------------------
int dil_shake256_dev(uint8_t* out, size_t out_bytes, const uint8_t* in, size_t in_bytes, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_shake256(reinterpret_cast<uint64_t*>(out), (int)out_bytes, reinterpret_cast<const uint64_t*>(in),
                                     (int)in_bytes, batch, S(stream));
}
int dil_expand_a_dev(int32_t* A, const uint8_t* rho, int level, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_expand_a(A, rho, 32, level, batch, S(stream));
}
int dil_expand_mask_dev(int32_t* y, const uint8_t* rhoprime, const uint32_t* kappa, int level, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_expand_mask(y, rhoprime, kappa, level, batch, S(stream));
}
int dil_sample_in_ball_dev(int32_t* c, const uint8_t* ctilde, int level, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_sample_in_ball(c, ctilde, level, batch, S(stream));
}
int dil_pack_w1_dev(uint8_t* out, const uint8_t* w1, int level, size_t batch, void* stream)
{
    int rc = ensure_init();
    if (rc) return rc;
    return (int)dil::launch_pack_w1(out, w1, level, batch, g.t, S(stream));
}

// ---- row N3 (first step): composite sequences, device-resident end to end -----------------------
namespace {
struct StreamScratch {       // stream-ordered temporaries, freed on the same stream
    hipStream_t s;
    void* p[40];
    int n = 0;
    explicit StreamScratch(hipStream_t st) : s(st) {}
    int get(void** out, size_t bytes)
    {
        hipError_t e = hipMallocAsync(out, bytes, s);
        if (e != hipSuccess) return (int)e;
        p[n++] = *out;
        return 0;
    }
    ~StreamScratch()
    {
        for (int i = 0; i < n; i++) (void)hipFreeAsync(p[i], s);
    }
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
    int rc = ensure_init(), K, L;
    if (rc || (rc = level_kl(level, &K, &L))) return rc;
    if (batch == 0) return 0;
    hipStream_t s = S(stream);
    StreamScratch ws(s);
    void *c, *w1, *w1p;
    const size_t wb = (size_t)K * (level == 2 ? 192 : 128);
    if ((rc = ws.get(&c, batch * 1024)) || (rc = ws.get(&w1, batch * K * 256)) || (rc = ws.get(&w1p, batch * wb))) return rc;
    DIL_TRY(dil::launch_z_norm(verdict, z, level, batch, s));
    DIL_TRY(dil::launch_sample_in_ball(static_cast<int32_t*>(c), ctilde, level, batch, s));
    DIL_TRY(dil::launch_verify(level, static_cast<uint8_t*>(w1), A, z, static_cast<int32_t*>(c), t1, h, batch, shared_pk, g.t, s));
    DIL_TRY(dil::launch_pack_w1(static_cast<uint8_t*>(w1p), static_cast<uint8_t*>(w1), level, batch, g.t, s));
    DIL_TRY(dil::launch_challenge_hash(nullptr, verdict, mu, static_cast<uint8_t*>(w1p), level, ctilde, batch, s));
    return 0;
}

namespace {
// temporaries of one signing attempt over `batch` entries
struct AttemptScratch {
    int32_t* y; uint8_t* w1; int32_t* w0; uint8_t* w1p; int32_t* c;
    int alloc(StreamScratch& ws, int level, int K, int L, size_t batch)
    {
        void *py, *pw1, *pw0, *pw1p, *pc;
        int rc;
        if ((rc = ws.get(&py, batch * L * 1024)) || (rc = ws.get(&pw1, batch * K * 256)) || (rc = ws.get(&pw0, batch * K * 1024)) ||
            (rc = ws.get(&pw1p, batch * K * (level == 2 ? 192 : 128))) || (rc = ws.get(&pc, batch * 1024)))
            return rc;
        y = static_cast<int32_t*>(py); w1 = static_cast<uint8_t*>(pw1); w0 = static_cast<int32_t*>(pw0);
        w1p = static_cast<uint8_t*>(pw1p); c = static_cast<int32_t*>(pc);
        return 0;
    }
};
int sign_attempt_impl(const AttemptScratch& t, uint8_t* ctilde, int32_t* z, uint8_t* h, int32_t* flags, const int32_t* A,
                      const uint8_t* mu, const uint8_t* rhoprime, const uint32_t* kappa, const int32_t* s1hat, const int32_t* s2hat,
                      const int32_t* t0hat, int level, size_t batch, int shared_key, hipStream_t s, dil::KeyMap km = dil::KeyMap(),
                      int phases = 3)
{
    if (phases & 1) DIL_TRY(dil::launch_expand_mask(t.y, rhoprime, kappa, level, batch, s));
    if (!(phases & 2)) return 0;
    DIL_TRY(dil::launch_matvec(level, dil::OUT_W1W0, nullptr, t.w1, t.w0, A, t.y, batch, shared_key, g.t, s, km));
    DIL_TRY(dil::launch_pack_w1(t.w1p, t.w1, level, batch, g.t, s));
    DIL_TRY(dil::launch_challenge_hash(ctilde, nullptr, mu, t.w1p, level, nullptr, batch, s));
    DIL_TRY(dil::launch_sample_in_ball(t.c, ctilde, level, batch, s));
    DIL_TRY(dil::launch_sign2(level, z, h, flags, t.c, t.y, t.w0, t.w1, s1hat, s2hat, t0hat, batch, shared_key, g.t, s, km));
    return 0;
}

// The same attempt over entries [off, off + cnt) of the per-entry arrays (per-key arrays are addressed through km)
int sign_attempt_range(const AttemptScratch& t, uint8_t* ctilde, int32_t* z, uint8_t* h, int32_t* flags, const int32_t* A,
                       const uint8_t* mu, const uint8_t* rhoprime, const uint32_t* kappa, const int32_t* s1hat, const int32_t* s2hat,
                       const int32_t* t0hat, int level, int K, int L, size_t off, size_t cnt, int shared_key, hipStream_t s, dil::KeyMap km,
                       int phases = 3)
{
    AttemptScratch u = t;
    u.y += off * L * 256; u.w1 += off * K * 256; u.w0 += off * K * 256; u.w1p += off * K * (level == 2 ? 192 : 128); u.c += off * 256;
    km.base += (uint32_t)off;
    return sign_attempt_impl(u, ctilde + off * 32, z + off * L * 256, h + off * K * 256, flags + off, A, mu + off * 64,
                             rhoprime + off * 64, kappa + off, s1hat, s2hat, t0hat, level, cnt, shared_key, s, km, phases);
}

// A second stream for the signing loop: the hash kernels of a round are latency-bound (one sponge per lane, a few
// hundred waves), the polynomial kernels throughput-bound; two half-rounds in flight overlap the two kinds.
struct AuxStream {
    std::mutex mu;
    hipStream_t s = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;   // fork: the aux half may start; join: it is done
    int device = -1;
    bool ensure(int dev)
    {
        if (s && device == dev) return true;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return false;
        if (hipEventCreateWithFlags(&join, hipEventDisableTiming) != hipSuccess) return false;
        device = dev;
        return true;
    }
};
AuxStream g_aux;

// Run an independent part of a composite call on the helper stream (if nobody else is using it): fork() returns the
// stream to launch that part on -- the helper, ordered after everything already on `main`, or `main` itself -- and
// join() makes `main` wait for it.
struct AuxFork {
    std::unique_lock<std::mutex> lk;
    hipStream_t main;
    bool on, forked = false;
    explicit AuxFork(hipStream_t m) : lk(g_aux.mu, std::try_to_lock), main(m)
    {
        on = g.aux_overlap && lk.owns_lock() && g_aux.ensure(g.device);
    }
    // `sponges`: lanes of the lane-per-sponge work going to the helper.  Only latency-bound work (less than about one
    // wave per SIMD) gains from running beside the main stream; throughput-bound work just pays the fork/join.
    hipStream_t fork(size_t sponges)
    {
        if (!on || sponges >= (size_t)g.t.num_cus * 256) return main;
        if (hipEventRecord(g_aux.fork, main) != hipSuccess || hipStreamWaitEvent(g_aux.s, g_aux.fork, 0) != hipSuccess) {
            on = false;
            return main;
        }
        forked = true;
        return g_aux.s;
    }
    int join()
    {
        if (!forked) return 0;
        forked = false;
        DIL_TRY(hipEventRecord(g_aux.join, g_aux.s));
        DIL_TRY(hipStreamWaitEvent(main, g_aux.join, 0));
        return 0;
    }
    ~AuxFork() { (void)join(); }
};
}  // namespace

int dil_sign_attempt_dev(uint8_t* ctilde, int32_t* z, uint8_t* h, int32_t* flags, const int32_t* A, const uint8_t* mu,
                         const uint8_t* rhoprime, const uint32_t* kappa, const int32_t* s1hat, const int32_t* s2hat,
                         const int32_t* t0hat, int level, size_t batch, int shared_key, void* stream)
{
    int rc = ensure_init(), K, L;
    if (rc || (rc = level_kl(level, &K, &L))) return rc;
    if (batch == 0) return 0;
    hipStream_t s = S(stream);
    StreamScratch ws(s);
    AttemptScratch t;
    if ((rc = t.alloc(ws, level, K, L, batch))) return rc;
    return sign_attempt_impl(t, ctilde, z, h, flags, A, mu, rhoprime, kappa, s1hat, s2hat, t0hat, level, batch, shared_key, s);
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
    int rc = ensure_init();
    if (rc || (rc = level_par(level, &p)) || (rc = codec_desc(kind, p, &d))) return rc;
    return (int)dil::launch_unpack(d.bits, out, in, in_stride, in_offset, d.polys, d.xf, d.offset, batch, g.t, S(stream));
}
int dil_pack_dev(uint8_t* out, size_t out_stride, size_t out_offset, const int32_t* in, int kind, int level, size_t batch, void* stream)
{
    LevelPar p;
    CodecDesc d;
    int rc = ensure_init();
    if (rc || (rc = level_par(level, &p)) || (rc = codec_desc(kind, p, &d))) return rc;
    return (int)dil::launch_pack(d.bits, out, out_stride, out_offset, in, d.polys, d.xf, d.offset, batch, g.t, S(stream));
}
int dil_hint_unpack_dev(uint8_t* h, int32_t* bad, const uint8_t* in, size_t in_stride, size_t in_offset, int level, size_t batch, void* stream)
{
    LevelPar p;
    int rc = ensure_init();
    if (rc || (rc = level_par(level, &p))) return rc;
    return (int)dil::launch_hint_unpack(h, bad, in, in_stride, in_offset, p.K, p.omega, batch, S(stream));
}
int dil_hint_pack_dev(uint8_t* out, size_t out_stride, size_t out_offset, const uint8_t* h, int level, size_t batch, void* stream)
{
    LevelPar p;
    int rc = ensure_init();
    if (rc || (rc = level_par(level, &p))) return rc;
    return (int)dil::launch_hint_pack(out, out_stride, out_offset, h, p.K, p.omega, batch, S(stream));
}
int dil_expand_s_dev(int32_t* s1, int32_t* s2, const uint8_t* rhoprime, size_t stride, int level, size_t batch, void* stream)
{
    LevelPar p;
    int rc = ensure_init();
    if (rc || (rc = level_par(level, &p))) return rc;
    return (int)dil::launch_expand_s(s1, s2, rhoprime, stride, p.eta, p.L, p.K, batch, S(stream));
}

int dil_keygen_dev(uint8_t* pk, uint8_t* sk, const uint8_t* seed, int level, size_t batch, void* stream)
{
    LevelPar p;
    int rc = ensure_init();
    if (rc || (rc = level_par(level, &p))) return rc;
    if (batch == 0) return 0;
    hipStream_t s = S(stream);
    StreamScratch ws(s);
    const size_t pkb = dil_pk_bytes(level), skb = dil_sk_bytes(level), sb = (size_t)32 * p.eta_bits;
    void *exp, *A, *s1, *s2, *w, *t1, *t0, *tr;
    if ((rc = ws.get(&exp, batch * 128)) || (rc = ws.get(&A, batch * p.K * p.L * 1024)) || (rc = ws.get(&s1, batch * p.L * 1024)) ||
        (rc = ws.get(&s2, batch * p.K * 1024)) || (rc = ws.get(&w, batch * p.K * 1024)) || (rc = ws.get(&t1, batch * p.K * 1024)) ||
        (rc = ws.get(&t0, batch * p.K * 1024)) || (rc = ws.get(&tr, batch * 32)))
        return rc;
    uint8_t* e = static_cast<uint8_t*>(exp);                       // rho(32) | rho'(64) | key(32)  (KG_*)
    int32_t *s1p = static_cast<int32_t*>(s1), *s2p = static_cast<int32_t*>(s2);
    AuxFork ax(s);
    DIL_TRY(dil::launch_shake256(static_cast<uint64_t*>(exp), 128, reinterpret_cast<const uint64_t*>(seed), 32, batch, s));
    // ExpandS (helper stream) runs beside ExpandA: both are Keccak-bound and independent
    DIL_TRY(dil::launch_expand_s(s1p, s2p, e + 32, 128, p.eta, p.L, p.K, batch, ax.fork(batch * p.K * p.L)));
    DIL_TRY(dil::launch_expand_a(static_cast<int32_t*>(A), e, 128, level, batch, s));
    if ((rc = ax.join())) return rc;
    DIL_TRY(dil::launch_matvec(level, dil::OUT_W, static_cast<int32_t*>(w), nullptr, nullptr, static_cast<int32_t*>(A), s1p, batch, 0, g.t, s));
    DIL_TRY(dil::launch_power2round(static_cast<int32_t*>(t1), static_cast<int32_t*>(t0), static_cast<int32_t*>(w), s2p,
                                    batch * p.K * 256, g.t, s));
    // pk = rho | t1
    DIL_TRY(dil::launch_copy_field(pk, pkb, 0, e, 128, 0, 32, batch, g.t, s));
    DIL_TRY(dil::launch_pack(10, pk, pkb, 32, static_cast<int32_t*>(t1), p.K, dil::XF_PLAIN, 0, batch, g.t, s));
    // tr = SHAKE256(pk, 32) (pk length is a multiple of 8 at every level): one long sponge per key, latency-bound --
    // on the helper stream, under the packing of the rest of sk
    {
        hipStream_t a = ax.fork(batch);
        DIL_TRY(dil::launch_shake256(static_cast<uint64_t*>(tr), 32, reinterpret_cast<const uint64_t*>(pk), (int)pkb, batch, a));
        DIL_TRY(dil::launch_copy_field(sk, skb, 64, static_cast<uint8_t*>(tr), 32, 0, 32, batch, g.t, a));
    }
    // sk = rho | key | tr | s1 | s2 | t0
    DIL_TRY(dil::launch_copy_field(sk, skb, 0, e, 128, 0, 32, batch, g.t, s));
    DIL_TRY(dil::launch_copy_field(sk, skb, 32, e, 128, 96, 32, batch, g.t, s));
    DIL_TRY(dil::launch_pack(p.eta_bits, sk, skb, 96, s1p, p.L, dil::XF_OFFSET_MINUS, p.eta, batch, g.t, s));
    DIL_TRY(dil::launch_pack(p.eta_bits, sk, skb, 96 + p.L * sb, s2p, p.K, dil::XF_OFFSET_MINUS, p.eta, batch, g.t, s));
    DIL_TRY(dil::launch_pack(13, sk, skb, 96 + (p.L + p.K) * sb, static_cast<int32_t*>(t0), p.K, dil::XF_OFFSET_MINUS, 1 << 12, batch, g.t, s));
    return ax.join();
}

int dil_verify_sig_dev(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level, size_t batch,
                       int shared_pk, void* stream)
{
    LevelPar p;
    int rc = ensure_init();
    if (rc || (rc = level_par(level, &p))) return rc;
    if (batch == 0) return 0;
    if (reinterpret_cast<uintptr_t>(pk) & 7) return (int)hipErrorInvalidValue;      // rho is read as 64-bit words
    hipStream_t s = S(stream);
    StreamScratch ws(s);
    const size_t pkb = dil_pk_bytes(level), sgb = dil_sig_bytes(level), zb = (size_t)p.L * 32 * p.zbits;
    const size_t nk = shared_pk ? 1 : batch, wb = (size_t)p.K * (level == 2 ? 192 : 128);
    void *A, *t1, *z, *h, *bad, *ct, *c, *w1, *w1p;
    if ((rc = ws.get(&A, nk * p.K * p.L * 1024)) || (rc = ws.get(&t1, nk * p.K * 1024)) || (rc = ws.get(&z, batch * p.L * 1024)) ||
        (rc = ws.get(&h, batch * p.K * 256)) || (rc = ws.get(&bad, batch * 4)) || (rc = ws.get(&ct, batch * 32)) ||
        (rc = ws.get(&c, batch * 1024)) || (rc = ws.get(&w1, batch * p.K * 256)) || (rc = ws.get(&w1p, batch * wb)))
        return rc;
    AuxFork ax(s);
    {   // public-key side (helper stream): A = ExpandA(rho), t1
        hipStream_t a = ax.fork(nk * p.K * p.L);
        DIL_TRY(dil::launch_expand_a(static_cast<int32_t*>(A), pk, pkb, level, nk, a));
        DIL_TRY(dil::launch_unpack(10, static_cast<int32_t*>(t1), pk, pkb, 32, p.K, dil::XF_PLAIN, 0, nk, g.t, a));
    }
    // signature side: c~, z, hints, ||z|| check, c = SampleInBall(c~)
    DIL_TRY(dil::launch_copy_field(static_cast<uint8_t*>(ct), 32, 0, sig, sgb, 0, 32, batch, g.t, s));
    DIL_TRY(dil::launch_unpack(p.zbits, static_cast<int32_t*>(z), sig, sgb, 32, p.L, dil::XF_OFFSET_MINUS, p.gamma1, batch, g.t, s));
    DIL_TRY(dil::launch_hint_unpack(static_cast<uint8_t*>(h), static_cast<int32_t*>(bad), sig, sgb, 32 + zb, p.K, p.omega, batch, s));
    DIL_TRY(dil::launch_z_norm(verdict, static_cast<int32_t*>(z), level, batch, s));
    DIL_TRY(dil::launch_sample_in_ball(static_cast<int32_t*>(c), static_cast<uint8_t*>(ct), level, batch, s));
    if ((rc = ax.join())) return rc;
    DIL_TRY(dil::launch_verify(level, static_cast<uint8_t*>(w1), static_cast<int32_t*>(A), static_cast<int32_t*>(z),
                               static_cast<int32_t*>(c), static_cast<int32_t*>(t1), static_cast<uint8_t*>(h), batch, shared_pk, g.t, s));
    DIL_TRY(dil::launch_pack_w1(static_cast<uint8_t*>(w1p), static_cast<uint8_t*>(w1), level, batch, g.t, s));
    DIL_TRY(dil::launch_challenge_hash(nullptr, verdict, mu, static_cast<uint8_t*>(w1p), level, static_cast<uint8_t*>(ct), batch, s));
    return (int)dil::launch_or_flag(verdict, static_cast<int32_t*>(bad), 4, batch, g.t, s);
}

// ---- row N3: the whole signing rejection loop on the device ---------------------------------------
// combined_top.v's sign FSMs (:1694-2229) retry one signature until it passes.  A batch engine is
// better used WIDE than deep: each round runs S speculative attempts (kappa = a0*L, (a0+1)*L, ...)
// for every still-pending signature, S chosen so that a round keeps about `cap` entries in flight;
// the first accepted attempt of an item wins, which is exactly the signature the sequential loop
// produces.  The pending set shrinks geometrically while S grows, so the loop needs ~5 rounds
// instead of the ~35 the unluckiest signature of a large batch takes.
int dil_sign_dev(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* mu, int level, size_t batch, int shared_sk,
                 int max_attempts, void* stream)
{
    LevelPar p;
    int rc = ensure_init();
    if (rc || (rc = level_par(level, &p))) return rc;
    if (batch == 0) return 0;
    if (batch > 0x3fffffffull || max_attempts <= 0) return (int)hipErrorInvalidValue;
    if (batch == 1) shared_sk = 1;
    hipStream_t s = S(stream);
    StreamScratch ws(s);
    const size_t skb = dil_sk_bytes(level), sgb = dil_sig_bytes(level), sb = (size_t)32 * p.eta_bits, zb = (size_t)p.L * 32 * p.zbits;
    const size_t nk = shared_sk ? 1 : batch, sk_stride = shared_sk ? 0 : skb;
    const size_t a_row = (size_t)p.K * p.L * 1024, l_row = (size_t)p.L * 1024, k_row = (size_t)p.K * 1024;
    // entries kept in flight per round: the hash kernels are latency-bound below ~1 wave per SIMD, so small batches
    // speculate for free
    const size_t cap = std::max<size_t>(batch, g.sign_cap ? (size_t)g.sign_cap : 16384);
    const int s_max = 64;
    void *A, *s1h, *s2h, *t0h, *km, *rp, *idx0, *idx1, *cnt, *kap, *ct, *z, *h, *fl, *wine, *wini, *mu_c, *rp_c;
    AttemptScratch att;
    if ((rc = ws.get(&A, nk * a_row)) || (rc = ws.get(&s1h, nk * l_row)) || (rc = ws.get(&s2h, nk * k_row)) ||
        (rc = ws.get(&t0h, nk * k_row)) || (rc = ws.get(&km, batch * 96)) || (rc = ws.get(&rp, batch * 64)) ||
        (rc = ws.get(&idx0, batch * 4)) || (rc = ws.get(&idx1, batch * 4)) || (rc = ws.get(&cnt, 8)) || (rc = ws.get(&kap, cap * 4)) ||
        (rc = ws.get(&ct, cap * 32)) || (rc = ws.get(&z, cap * l_row)) || (rc = ws.get(&h, cap * p.K * 256)) ||
        (rc = ws.get(&fl, cap * 4)) || (rc = ws.get(&wine, batch * 4)) || (rc = ws.get(&wini, batch * 4)) ||
        (rc = ws.get(&mu_c, cap * 64)) || (rc = ws.get(&rp_c, cap * 64)) || (rc = att.alloc(ws, level, p.K, p.L, cap)))
        return rc;
    // key material: A = ExpandA(rho), s1^ s2^ t0^ = NTT(unpack(sk))
    DIL_TRY(dil::launch_expand_a(static_cast<int32_t*>(A), sk, skb, level, nk, s));
    DIL_TRY(dil::launch_unpack(p.eta_bits, static_cast<int32_t*>(s1h), sk, skb, 96, p.L, dil::XF_OFFSET_MINUS, p.eta, nk, g.t, s));
    DIL_TRY(dil::launch_unpack(p.eta_bits, static_cast<int32_t*>(s2h), sk, skb, 96 + p.L * sb, p.K, dil::XF_OFFSET_MINUS, p.eta, nk, g.t, s));
    DIL_TRY(dil::launch_unpack(13, static_cast<int32_t*>(t0h), sk, skb, 96 + (p.L + p.K) * sb, p.K, dil::XF_OFFSET_MINUS, 1 << 12, nk, g.t, s));
    DIL_TRY(dil::launch_ntt(false, dil::LAYOUT_POLY, 0, static_cast<int32_t*>(s1h), nk * p.L, g.t, s));
    DIL_TRY(dil::launch_ntt(false, dil::LAYOUT_POLY, 0, static_cast<int32_t*>(s2h), nk * p.K, g.t, s));
    DIL_TRY(dil::launch_ntt(false, dil::LAYOUT_POLY, 0, static_cast<int32_t*>(t0h), nk * p.K, g.t, s));
    // rho' = SHAKE256(key || mu, 64)  (deterministic signing, as the reference's KATs)
    DIL_TRY(dil::launch_copy_field(static_cast<uint8_t*>(km), 96, 0, sk, sk_stride, 32, 32, batch, g.t, s));
    DIL_TRY(dil::launch_copy_field(static_cast<uint8_t*>(km), 96, 32, mu, 64, 0, 64, batch, g.t, s));
    DIL_TRY(dil::launch_shake256(static_cast<uint64_t*>(rp), 64, static_cast<uint64_t*>(km), 96, batch, s));
    DIL_TRY(hipMemsetAsync(attempts, 0, batch * 4, s));

    std::unique_lock<std::mutex> aux_lock(g_aux.mu, std::try_to_lock);   // one signing loop at a time uses the aux stream
    const bool two_streams = g.sign_streams > 1 && aux_lock.owns_lock() && g_aux.ensure(g.device);
    int32_t *idx_cur = nullptr, *idx_next = static_cast<int32_t*>(idx0);
    size_t n = batch;
    int a0 = 0;                                          // attempts every pending item has already failed
    while (n > 0 && a0 < max_attempts) {
        const int S_ = (int)std::min<size_t>(std::min<size_t>(cap / n, (size_t)s_max), (size_t)(max_attempts - a0));
        const size_t E = n * (size_t)S_;
        const bool direct = !idx_cur && S_ == 1;         // first round of a full batch: the caller's arrays as they are
        const uint8_t *mur = mu, *rpr = static_cast<uint8_t*>(rp);
        if (!direct) {
            DIL_TRY(dil::launch_gather_rows(mu_c, mu, idx_cur, 64, (uint32_t)S_, E, g.t, s));
            DIL_TRY(dil::launch_gather_rows(rp_c, rp, idx_cur, 64, (uint32_t)S_, E, g.t, s));
            mur = static_cast<uint8_t*>(mu_c);
            rpr = static_cast<uint8_t*>(rp_c);
        }
        dil::KeyMap keys;                                // per-item keys are read in place through the pending list
        keys.idx = idx_cur;
        keys.S = (uint32_t)S_;
        DIL_TRY(dil::launch_sign_kappa(static_cast<uint32_t*>(kap), (uint32_t)a0, (uint32_t)p.L, (uint32_t)S_, E, s));
        // Two half-rounds, staggered by one kernel: the aux half starts when the main half's ExpandMask is done, so its
        // throughput-bound kernels run under the main half's latency-bound hashing and vice versa.
        const size_t half = two_streams && E >= 4096 ? (E / 2) : 0;      // entries [half, E) on the aux stream
        auto part = [&](size_t off, size_t cnt, hipStream_t st, int phases) {
            return sign_attempt_range(att, static_cast<uint8_t*>(ct), static_cast<int32_t*>(z), static_cast<uint8_t*>(h),
                                      static_cast<int32_t*>(fl), static_cast<int32_t*>(A), mur, rpr, static_cast<uint32_t*>(kap),
                                      static_cast<int32_t*>(s1h), static_cast<int32_t*>(s2h), static_cast<int32_t*>(t0h), level, p.K, p.L,
                                      off, cnt, shared_sk, st, keys, phases);
        };
        if (half) {
            if ((rc = part(0, half, s, 1))) return rc;                   // main: ExpandMask
            DIL_TRY(hipEventRecord(g_aux.fork, s));
            DIL_TRY(hipStreamWaitEvent(g_aux.s, g_aux.fork, 0));
            if ((rc = part(half, E - half, g_aux.s, 3))) return rc;      // aux: the whole chain
            if ((rc = part(0, half, s, 2))) return rc;                   // main: the rest
            DIL_TRY(hipEventRecord(g_aux.join, g_aux.s));
            DIL_TRY(hipStreamWaitEvent(s, g_aux.join, 0));
        } else if ((rc = part(0, E, s, 3))) {
            return rc;
        }
        // winners (first accepted attempt per item) -> packed straight into their signature slots
        int32_t* counts = static_cast<int32_t*>(cnt);
        DIL_TRY(hipMemsetAsync(cnt, 0, 8, s));
        DIL_TRY(dil::launch_sign_collect(attempts, idx_next, static_cast<int32_t*>(wine), static_cast<int32_t*>(wini), counts,
                                         static_cast<int32_t*>(fl), idx_cur, a0, S_, n, s));
        dil::RowMap win;
        win.src_row = static_cast<int32_t*>(wine);
        win.dst_row = static_cast<int32_t*>(wini);
        win.count = counts + 1;
        DIL_TRY(dil::launch_copy_field(sig, sgb, 0, static_cast<uint8_t*>(ct), 32, 0, 32, n, g.t, s, win));
        DIL_TRY(dil::launch_pack(p.zbits, sig, sgb, 32, static_cast<int32_t*>(z), p.L, dil::XF_OFFSET_MINUS, p.gamma1, n, g.t, s, win));
        DIL_TRY(dil::launch_hint_pack(sig, sgb, 32 + zb, static_cast<uint8_t*>(h), p.K, p.omega, n, s, win));
        int32_t pending = 0;
        DIL_TRY(hipMemcpyAsync(&pending, cnt, 4, hipMemcpyDeviceToHost, s));
        DIL_TRY(hipStreamSynchronize(s));
        n = (size_t)pending;
        a0 += S_;
        idx_cur = idx_next;
        idx_next = idx_cur == static_cast<int32_t*>(idx0) ? static_cast<int32_t*>(idx1) : static_cast<int32_t*>(idx0);
    }
    return n == 0 ? 0 : DIL_ERR_UNFINISHED;
}

// ---- host-buffer forms of the whole operations (H2D -> device call -> D2H on the null stream) --------
namespace {
struct DevBuf {
    void* p = nullptr;
    int alloc(size_t bytes) { return (int)hipMalloc(&p, bytes ? bytes : 1); }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
}  // namespace

int dil_keygen_host(uint8_t* pk, uint8_t* sk, const uint8_t* seed, int level, size_t batch)
{
    const size_t pkb = dil_pk_bytes(level), skb = dil_sk_bytes(level);
    if (!pkb) return (int)hipErrorInvalidValue;
    if (batch == 0) return 0;
    int rc = ensure_init();
    if (rc) return rc;
    DevBuf dpk, dsk, dseed;
    if ((rc = dpk.alloc(batch * pkb)) || (rc = dsk.alloc(batch * skb)) || (rc = dseed.alloc(batch * 32))) return rc;
    DIL_TRY(hipMemcpy(dseed.p, seed, batch * 32, hipMemcpyHostToDevice));
    rc = dil_keygen_dev(static_cast<uint8_t*>(dpk.p), static_cast<uint8_t*>(dsk.p), static_cast<uint8_t*>(dseed.p), level, batch, nullptr);
    if (rc) return rc;
    DIL_TRY(hipMemcpy(pk, dpk.p, batch * pkb, hipMemcpyDeviceToHost));
    DIL_TRY(hipMemcpy(sk, dsk.p, batch * skb, hipMemcpyDeviceToHost));
    return 0;
}

int dil_sign_host(uint8_t* sig, int32_t* attempts, const uint8_t* sk, const uint8_t* mu, int level, size_t batch, int shared_sk,
                  int max_attempts)
{
    const size_t skb = dil_sk_bytes(level), sgb = dil_sig_bytes(level);
    if (!skb) return (int)hipErrorInvalidValue;
    if (batch == 0) return 0;
    int rc = ensure_init();
    if (rc) return rc;
    const size_t nk = shared_sk ? 1 : batch;
    DevBuf dsig, datt, dsk, dmu;
    if ((rc = dsig.alloc(batch * sgb)) || (rc = datt.alloc(batch * 4)) || (rc = dsk.alloc(nk * skb)) || (rc = dmu.alloc(batch * 64)))
        return rc;
    DIL_TRY(hipMemcpy(dsk.p, sk, nk * skb, hipMemcpyHostToDevice));
    DIL_TRY(hipMemcpy(dmu.p, mu, batch * 64, hipMemcpyHostToDevice));
    const int src = dil_sign_dev(static_cast<uint8_t*>(dsig.p), static_cast<int32_t*>(datt.p), static_cast<uint8_t*>(dsk.p),
                                 static_cast<uint8_t*>(dmu.p), level, batch, shared_sk, max_attempts, nullptr);
    if (src && src != DIL_ERR_UNFINISHED) return src;
    DIL_TRY(hipDeviceSynchronize());
    DIL_TRY(hipMemcpy(sig, dsig.p, batch * sgb, hipMemcpyDeviceToHost));
    if (attempts) DIL_TRY(hipMemcpy(attempts, datt.p, batch * 4, hipMemcpyDeviceToHost));
    return src;
}

int dil_verify_sig_host(int32_t* verdict, const uint8_t* pk, const uint8_t* sig, const uint8_t* mu, int level, size_t batch,
                        int shared_pk)
{
    const size_t pkb = dil_pk_bytes(level), sgb = dil_sig_bytes(level);
    if (!pkb) return (int)hipErrorInvalidValue;
    if (batch == 0) return 0;
    int rc = ensure_init();
    if (rc) return rc;
    const size_t nk = shared_pk ? 1 : batch;
    DevBuf dv, dpk, dsig, dmu;
    if ((rc = dv.alloc(batch * 4)) || (rc = dpk.alloc(nk * pkb)) || (rc = dsig.alloc(batch * sgb)) || (rc = dmu.alloc(batch * 64)))
        return rc;
    DIL_TRY(hipMemcpy(dpk.p, pk, nk * pkb, hipMemcpyHostToDevice));
    DIL_TRY(hipMemcpy(dsig.p, sig, batch * sgb, hipMemcpyHostToDevice));
    DIL_TRY(hipMemcpy(dmu.p, mu, batch * 64, hipMemcpyHostToDevice));
    rc = dil_verify_sig_dev(static_cast<int32_t*>(dv.p), static_cast<uint8_t*>(dpk.p), static_cast<uint8_t*>(dsig.p),
                            static_cast<uint8_t*>(dmu.p), level, batch, shared_pk, nullptr);
    if (rc) return rc;
    DIL_TRY(hipMemcpy(verdict, dv.p, batch * 4, hipMemcpyDeviceToHost));
    return 0;
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
