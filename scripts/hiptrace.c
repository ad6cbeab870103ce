/* scripts/hiptrace.c -- diagnosis tool (LD_PRELOAD), not product: a ring buffer of the process's last HIP memory calls (copies, host
 * registrations, allocations, frees, kernel launches: kind, pointers, size, stream, thread, time) that costs ~50 ns a call, dumped together
 * with /proc/self/maps and a native backtrace when the process gets SIGABRT / SIGSEGV (ROCr's "Memory access fault by GPU" handler calls
 * abort() on its event thread).  With the fault address ROCr prints, the dump says which call handed that address to the GPU and whether
 * the range is still mapped.   gcc -O2 -shared -fPIC -o hiptrace.so hiptrace.c -ldl -lpthread ; HIPTRACE_OUT=/path LD_PRELOAD=hiptrace.so python ...
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#ifdef HIPTRACE_FULL
#define RING (1 << 20)      /* 64 MiB of records: the whole life of a test-suite process (with it, and the madvise hook, no run of 11 died: r06e, r06f) */
#else
#define RING 16384          /* the form under which runs did die (r06c: 1 of 4) */
#endif
typedef struct { uint64_t t_ns; const char* what; const void *a, *b; size_t n; const void* stream; int tid, rc; } Rec;
static Rec ring[RING];
static volatile uint64_t head;
static __thread int my_tid;

static inline uint64_t now_ns(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (uint64_t)ts.tv_sec * 1000000000ull + ts.tv_nsec; }
static inline Rec* put(const char* what, const void* a, const void* b, size_t n, const void* stream)
{
    if (!my_tid) my_tid = (int)syscall(SYS_gettid);
    Rec* r = &ring[__atomic_fetch_add(&head, 1, __ATOMIC_RELAXED) % RING];
    r->t_ns = now_ns(); r->what = what; r->a = a; r->b = b; r->n = n; r->stream = stream; r->tid = my_tid; r->rc = -1;
    return r;
}
/* the real entry point: torch brings its own libamdhip64.so in with a LOCAL dlopen, where RTLD_NEXT does not look -- find the loaded object by name */
#include <link.h>
static char hip_path[512];
static int find_hip(struct dl_phdr_info* info, size_t size, void* data)
{
    (void)size; (void)data;
    if (info->dlpi_name && strstr(info->dlpi_name, "libamdhip64")) { strncpy(hip_path, info->dlpi_name, sizeof hip_path - 1); return 1; }
    return 0;
}
static void* resolve(const char* name)
{
    void* p = dlsym(RTLD_NEXT, name);
    if (p) return p;
    static void* h;
    if (!h) {
        dl_iterate_phdr(find_hip, 0);
        h = dlopen(hip_path[0] ? hip_path : "libamdhip64.so", RTLD_NOW | (hip_path[0] ? RTLD_NOLOAD : 0));
    }
    p = h ? dlsym(h, name) : 0;
    if (!p) { fprintf(stderr, "hiptrace: cannot resolve %s\n", name); abort(); }
    return p;
}
#define REAL(name) static int (*real)(); if (!real) real = (int (*)())resolve(#name)

int hipMemcpy(void* d, const void* s, size_t n, int kind) { REAL(hipMemcpy); Rec* r = put(kind == 1 ? "hipMemcpy H2D" : kind == 2 ? "hipMemcpy D2H" : "hipMemcpy", d, s, n, 0); return r->rc = real(d, s, n, kind); }
int hipMemcpyAsync(void* d, const void* s, size_t n, int kind, void* st) { REAL(hipMemcpyAsync); Rec* r = put(kind == 1 ? "hipMemcpyAsync H2D" : kind == 2 ? "hipMemcpyAsync D2H" : "hipMemcpyAsync", d, s, n, st); return r->rc = real(d, s, n, kind, st); }
int hipMemcpyWithStream(void* d, const void* s, size_t n, int kind, void* st) { REAL(hipMemcpyWithStream); Rec* r = put(kind == 1 ? "hipMemcpyWithStream H2D" : kind == 2 ? "hipMemcpyWithStream D2H" : "hipMemcpyWithStream", d, s, n, st); return r->rc = real(d, s, n, kind, st); }
int hipMemcpyHtoD(void* d, void* s, size_t n) { REAL(hipMemcpyHtoD); Rec* r = put("hipMemcpyHtoD", d, s, n, 0); return r->rc = real(d, s, n); }
int hipMemcpyDtoH(void* d, void* s, size_t n) { REAL(hipMemcpyDtoH); Rec* r = put("hipMemcpyDtoH", d, s, n, 0); return r->rc = real(d, s, n); }
int hipMemcpyHtoDAsync(void* d, void* s, size_t n, void* st) { REAL(hipMemcpyHtoDAsync); Rec* r = put("hipMemcpyHtoDAsync", d, s, n, st); return r->rc = real(d, s, n, st); }
int hipMemcpyDtoHAsync(void* d, void* s, size_t n, void* st) { REAL(hipMemcpyDtoHAsync); Rec* r = put("hipMemcpyDtoHAsync", d, s, n, st); return r->rc = real(d, s, n, st); }
int hipHostRegister(void* p, size_t n, unsigned f) { REAL(hipHostRegister); Rec* r = put("hipHostRegister", p, 0, n, 0); return r->rc = real(p, n, f); }
int hipHostUnregister(void* p) { REAL(hipHostUnregister); Rec* r = put("hipHostUnregister", p, 0, 0, 0); return r->rc = real(p); }
int hipHostMalloc(void** p, size_t n, unsigned f) { REAL(hipHostMalloc); int rc = real(p, n, f); Rec* r = put("hipHostMalloc", p ? *p : 0, 0, n, 0); return r->rc = rc; }
int hipHostFree(void* p) { REAL(hipHostFree); Rec* r = put("hipHostFree", p, 0, 0, 0); return r->rc = real(p); }
int hipMalloc(void** p, size_t n) { REAL(hipMalloc); int rc = real(p, n); Rec* r = put("hipMalloc", p ? *p : 0, 0, n, 0); return r->rc = rc; }
int hipFree(void* p) { REAL(hipFree); Rec* r = put("hipFree", p, 0, 0, 0); return r->rc = real(p); }
int hipMallocAsync(void** p, size_t n, void* st) { REAL(hipMallocAsync); int rc = real(p, n, st); Rec* r = put("hipMallocAsync", p ? *p : 0, 0, n, st); return r->rc = rc; }
int hipMallocFromPoolAsync(void** p, size_t n, void* pool, void* st) { REAL(hipMallocFromPoolAsync); int rc = real(p, n, pool, st); Rec* r = put("hipMallocFromPoolAsync", p ? *p : 0, pool, n, st); return r->rc = rc; }
int hipFreeAsync(void* p, void* st) { REAL(hipFreeAsync); Rec* r = put("hipFreeAsync", p, 0, 0, st); return r->rc = real(p, st); }
int hipMemsetAsync(void* p, int v, size_t n, void* st) { REAL(hipMemsetAsync); Rec* r = put("hipMemsetAsync", p, 0, n, st); return r->rc = real(p, v, n, st); }
typedef struct { unsigned x, y, z; } dim3_;
int hipLaunchKernel(const void* f, dim3_ g, dim3_ b, void** args, size_t shm, void* st) { REAL(hipLaunchKernel); Rec* r = put("hipLaunchKernel", f, (void*)(uintptr_t)g.x, b.x, st); return r->rc = ((int (*)(const void*, dim3_, dim3_, void**, size_t, void*))real)(f, g, b, args, shm, st); }
int hipStreamSynchronize(void* st) { REAL(hipStreamSynchronize); Rec* r = put("hipStreamSynchronize", 0, 0, 0, st); return r->rc = real(st); }
int hipDeviceSynchronize(void) { REAL(hipDeviceSynchronize); Rec* r = put("hipDeviceSynchronize", 0, 0, 0, 0); return r->rc = real(); }
int hipStreamDestroy(void* st) { REAL(hipStreamDestroy); Rec* r = put("hipStreamDestroy", 0, 0, 0, st); return r->rc = real(st); }
int hipStreamCreateWithFlags(void** st, unsigned f) { REAL(hipStreamCreateWithFlags); int rc = real(st, f); Rec* r = put("hipStreamCreateWithFlags", 0, 0, f, st ? *st : 0); return r->rc = rc; }

/* the thunk marks every host range it registers with the driver (a runtime-internal pin of a pageable copy, or hipHostRegister) MADV_DONTFORK and
 * gives it back with MADV_DOFORK: these two calls ARE the life of a userptr registration, page-aligned range included */
#ifdef HIPTRACE_FULL
int madvise(void* addr, size_t len, int advice)
{
    static int (*real_madvise)(void*, size_t, int);
    if (!real_madvise) real_madvise = (int (*)(void*, size_t, int))dlsym(RTLD_NEXT, "madvise");
    if (advice == 10 || advice == 11) { Rec* r = put(advice == 10 ? "madvise DONTFORK" : "madvise DOFORK", addr, 0, len, 0); return r->rc = real_madvise(addr, len, advice); }
    return real_madvise(addr, len, advice);
}
pid_t fork(void)
{
    static pid_t (*real_fork)(void);
    if (!real_fork) real_fork = (pid_t (*)(void))dlsym(RTLD_NEXT, "fork");
    Rec* r = put("fork", 0, 0, 0, 0);
    const pid_t p = real_fork();
    if (p != 0) r->rc = (int)p;
    return p;
}

#endif

static int in_heap(const void* p, const uintptr_t (*heaps)[2], int nheaps)
{
    for (int i = 0; i < nheaps; i++)
        if ((uintptr_t)p >= heaps[i][0] && (uintptr_t)p < heaps[i][1]) return 1;
    return 0;
}

static void dump(int sig, siginfo_t* si)
{
    const char* dir = getenv("HIPTRACE_OUT");
    char path[512];
    snprintf(path, sizeof path, "%s/hiptrace_%d.txt", dir ? dir : "/tmp", (int)getpid());
    FILE* f = fopen(path, "w");
    if (!f) return;
    fprintf(f, "signal %d in tid %ld (si_code %d, si_addr %p), pid %d, now %.6f s\n", sig, (long)syscall(SYS_gettid), si ? si->si_code : 0, si ? si->si_addr : 0,
            (int)getpid(), now_ns() * 1e-9);
    void* bt[64];
    int n = backtrace(bt, 64);
    fflush(f);
    backtrace_symbols_fd(bt, n, fileno(f));
    fprintf(f, "---- /proc/self/maps ----\n");
    FILE* m = fopen("/proc/self/maps", "r");
    static uintptr_t heaps[256][2];
    int nheaps = 0;
    if (m) {
        char ln[512];
        while (fgets(ln, sizeof ln, m)) {
            fputs(ln, f);
            unsigned long lo_, hi_;
            if (strstr(ln, "[heap]") && nheaps < 256 && sscanf(ln, "%lx-%lx", &lo_, &hi_) == 2) { heaps[nheaps][0] = lo_; heaps[nheaps][1] = hi_; nheaps++; }
        }
        fclose(m);
    }
    uint64_t h = head, lo = h > RING ? h - RING : 0;
    fprintf(f, "---- of the last %llu of %llu calls: every one that touches the [heap], every registration / madvise / fork / stream event, and the last 3000 of any kind "
               "(t [s], tid, call, a, b, n, stream, rc) ----\n", (unsigned long long)(h - lo), (unsigned long long)h);
    for (uint64_t i = lo; i < h; i++) {
        Rec* r = &ring[i % RING];
        const int keep = i + 3000 >= h || in_heap(r->a, heaps, nheaps) || in_heap(r->b, heaps, nheaps) ||
                         (r->what && (strstr(r->what, "Register") || strstr(r->what, "madvise") || strstr(r->what, "fork") || strstr(r->what, "hipStreamCreate") ||
                                      strstr(r->what, "hipStreamDestroy") || strstr(r->what, "Host")));
        if (!keep) continue;
        Dl_info di;
        const char* sym = "";
        if (r->what && !strcmp(r->what, "hipLaunchKernel") && dladdr(r->a, &di) && di.dli_sname) sym = di.dli_sname;
        fprintf(f, "%.6f %d %s %p %p %zu %p %d %s\n", r->t_ns * 1e-9, r->tid, r->what ? r->what : "?", r->a, r->b, r->n, r->stream, r->rc, sym);
    }
    fclose(f);
}
static volatile int in_handler;
static void on_signal(int sig, siginfo_t* si, void* uc)
{
    (void)uc;
    if (!__atomic_exchange_n(&in_handler, 1, __ATOMIC_SEQ_CST)) dump(sig, si);
    signal(sig, SIG_DFL);
    raise(sig);
}
static int (*real_sigaction)(int, const struct sigaction*, struct sigaction*);
static void install(void)
{
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_signal;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
    real_sigaction(SIGABRT, &sa, 0);
    real_sigaction(SIGSEGV, &sa, 0);
    real_sigaction(SIGBUS, &sa, 0);
}
int sigaction(int sig, const struct sigaction* act, struct sigaction* old)       /* nobody replaces the handler (faulthandler, torch) */
{
    if (!real_sigaction) real_sigaction = dlsym(RTLD_NEXT, "sigaction");
    if ((sig == SIGABRT || sig == SIGSEGV || sig == SIGBUS) && act) { if (old) memset(old, 0, sizeof *old); return 0; }
    return real_sigaction(sig, act, old);
}
__attribute__((constructor)) static void init(void)
{
    if (!real_sigaction) real_sigaction = dlsym(RTLD_NEXT, "sigaction");
    void* warm[4];
    backtrace(warm, 4);
    install();
}
