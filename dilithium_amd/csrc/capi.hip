// capi.hip -- the extern "C" boundary of libdil256.so (declared in include/dil256.h) and the
// host runtime behind it: twiddle-table construction, device selection, scratch management
// for the host-pointer entry points.  Host language is C++ because the reference's
// dilithium-256/ is C++; nothing here is a CPU fallback -- every arithmetic entry
// point launches a HIP kernel and returns the hipError_t if that is not possible.
#include "capi_internal.hpp"

#include <algorithm>
#include <mutex>
#include <stdlib.h>
#include <string.h>

namespace dil {
namespace rt {
State g;
int ensure_init()
{
    if (g.ready) return 0;
    return dil_init(-1);
}
}  // namespace rt
}  // namespace dil

namespace {
using dil::rt::g;
using dil::rt::ensure_init;
using dil::rt::S;

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
    if (g.ready) dil::rt::release_scratch();        // arenas belong to the previous device
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
    if (const char* e = getenv("DIL_SIGN_EARLY")) g.sign_early = atoi(e);
    if (const char* e = getenv("DIL_SIGN_WASTE")) g.sign_waste = atoi(e);
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
    dil::rt::release_scratch();
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
