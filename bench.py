#!/usr/bin/env python3
"""bench.py -- the BASELINE.json metric on MI355X.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[1] -- batched forward + inverse NTT,
n = 256, q = 8380417, batch = 65536 polynomials per GPU (64 MiB).  One STEP = one forward-NTT
launch + one inverse-NTT launch over one batch = 131072 transforms.  Inputs are resident in
HBM before the timed region; steps rotate over --rotate distinct batches (default 8 = 512 MiB,
twice the 256 MiB Infinity Cache) so that the number is an HBM-streaming number and not an
L3-resident one (the L3-resident rate is reported beside it as `llc_resident_value`).
metric value = transforms / second over all GPUs (weak scaling: per-GPU work fixed).

Also reported on the same JSON line:
  roofline      dominant kernels (forward / inverse NTT, alternating, same bytes).  `frac` is the SINGLE-KERNEL
                fraction: algorithmic bytes (2048 B x 65536 per launch) / the average launch duration of
                the same K steps on ONE stream (two HIP events around that region, no events inside it:
                an event record between launches costs microseconds of GPU idle time) -- the duration
                rocprofv3 reports for the kernel.  `frac_overlapped` is the aggregate of the timed region
                (`value`), where --streams launches overlap.  peak 8 TB/s (MI355X_MICROARCH.md);
                traffic = HBM bytes per launch from the committed rocprofv3 PMC pass (`traffic_source`).
  cpu_baseline  the reference's own ntt()+invntt() (oracle/_ref, kind "reference") -- or our C
                restatement (kind "port") -- on one host core, bounded sample.
  secondary     Dilithium-3 verify cores / s (configs[3], batch 8192, distinct pk) with its own
                roofline fraction; and the final RCCL gather time when N > 1.
  end_to_end    SURVEY 8(d): the same two workloads through the HOST-pointer entry points (H2D + kernel + D2H), pageable and
                page-locked caller buffers, against the link's own copy rate measured in the run; never `value`.

Layout: `Bench` holds what every leg shares (library handle, streams, the event-timed `timed()` region, the clock probe);
one function per leg fills its part of the JSON line; main() strings them together.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK = 256 * 4 * 2.4e9 / 4    # wave64 VALU instructions/s at the DATA-SHEET clock: 256 CUs x 4 SIMDs, 2.4 GHz, 4 cycles per instruction
#                                    (the sign leg also reports the fraction at the clock OBSERVED during its timed region)
NTT_BYTES = 2048               # 1 KiB read + 1 KiB written per transform
VERIFY3_BYTES = 45 * 1024 + 0  # z 5 + c 1 + t1 6 + A 30 KiB + h 1.5 + w1 1.5 KiB (distinct pk)
BATCH = 65536
VBATCH = 8192


def cpu_baseline(sample_polys=4096, target_s=10.0):
    """reference ntt()+invntt() on ONE host core, bounded to ~target_s seconds"""
    from oracle.oracle import Oracle, Reference, splitmix64_polys
    o = Oracle()
    a = splitmix64_polys(sample_polys, seed=1)
    if Reference.available():
        r = Reference()
        f_ntt, f_inv, kind = r.addr("ntt"), r.addr("invntt"), "reference"
    else:
        f_ntt, f_inv, kind = o.fn_addr("orc_ntt"), o.fn_addr("orc_invntt"), "port"
    buf = a.copy()
    t1 = o.time_poly_fn(f_ntt, buf, 1) + o.time_poly_fn(f_inv, buf, 1)      # calibrate
    reps = max(1, int(target_s / max(t1, 1e-6)))
    buf = a.copy()
    t = 0.0
    for _ in range(reps):        # alternate so values stay bounded like the GPU run
        t += o.time_poly_fn(f_ntt, buf, 1)
        t += o.time_poly_fn(f_inv, buf, 1)
    n = 2 * reps * sample_polys
    return {"value": n / t, "unit": "NTT/s", "cores": 1, "kind": kind,
            "sample": f"{reps} x (ntt + invntt) over {sample_polys} polynomials = {n} transforms in {t:.1f} s, "
                      f"1 thread of {os.cpu_count()} host CPUs"}


def cpu_quota():
    """(effective CPUs the container may use, source): the cgroup CPU quota when there is one -- sched_getaffinity / nproc report
    the VISIBLE hardware threads, which a CPU-limited lease cannot all run at once"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]                      # cgroup v2
        if q != "max":
            return float(q) / float(per), "cgroup v2 cpu.max"
        return None, "cgroup v2 cpu.max = max (no quota)"
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())                     # cgroup v1
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return q / per, "cgroup v1 cpu.cfs_quota_us"
        return None, "cgroup v1: no quota"
    except Exception:
        return None, "no cgroup CPU controller visible"


def cpu_baseline_all_threads(per_poly_s, target_s=5.0, sample_polys=1024):
    """the same reference ntt()+invntt() on every host hardware thread at once (the reference itself is single-threaded;
    SURVEY 8(d) asks for both figures).  ctypes drops the GIL during the foreign call, so plain threads run in parallel."""
    import threading
    from oracle.oracle import Oracle, Reference, splitmix64_polys
    o = Oracle()
    if Reference.available():
        r = Reference()
        f_ntt, f_inv = r.addr("ntt"), r.addr("invntt")
    else:
        f_ntt, f_inv = o.fn_addr("orc_ntt"), o.fn_addr("orc_invntt")
    nthreads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    chunk = max(1, int(0.05 / max(per_poly_s * sample_polys, 1e-6)))      # ~50 ms per foreign call
    bufs = [splitmix64_polys(sample_polys, seed=100 + i) for i in range(nthreads)]
    done = [0] * nthreads
    deadline = [0.0]

    def work(i):       # time-bounded: a cgroup CPU quota below the visible CPU count must not stretch the run
        buf, n = bufs[i], 0
        while time.perf_counter() < deadline[0]:
            o.time_poly_fn(f_ntt, buf, chunk)
            o.time_poly_fn(f_inv, buf, chunk)
            n += 2 * chunk * sample_polys
        done[i] = n

    ths = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter()
    deadline[0] = t0 + target_s
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    n = sum(done)
    quota, qsrc = cpu_quota()
    one_thread = 1.0 / per_poly_s
    return {"value": n / dt, "unit": "NTT/s", "threads": nthreads,
            "cores": round(quota, 2) if quota else nthreads,
            "cores_note": f"threads started = visible hardware threads = {nthreads}; CPU quota of this container: "
                          f"{'%.2f CPUs' % quota if quota else 'none'} ({qsrc}); measured parallel speed-up over one thread "
                          f"{n / dt / one_thread:.1f}x -- `cores` is the quota when there is one",
            "sample": f"{nthreads} threads, each alternating {chunk} x ntt / {chunk} x invntt over {sample_polys} polynomials "
                      f"until {target_s:.0f} s had passed = {n} transforms in {dt:.1f} s"}


def cpu_baseline_verify(target_s=5.0):
    from oracle.oracle import Oracle
    o = Oracle()
    A, z, c, t1, h = synth_verify(64, 3)
    t = o.time_verify_core(3, A, z, c, t1, h)
    reps = max(1, int(target_s / max(t, 1e-6)))
    tt = sum(o.time_verify_core(3, A, z, c, t1, h) for _ in range(reps))
    return {"value": 64 * reps / tt, "unit": "verify/s", "cores": 1, "kind": "port",
            "sample": f"{64 * reps} level-3 verify cores (oracle C restatement) in {tt:.1f} s, 1 thread"}


def cpu_baseline_verify_all_threads(target_s=4.0):
    """the oracle's level-3 verify core on every visible hardware thread at once (time-bounded), quota-aware like the NTT leg"""
    import threading
    from oracle.oracle import Oracle
    o = Oracle()
    nthreads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    ins = [synth_verify(16, 300 + i) for i in range(min(nthreads, 8))]
    done = [0] * nthreads
    deadline = [0.0]

    def work(i):
        A, z, c, t1, h = ins[i % len(ins)]
        n = 0
        while time.perf_counter() < deadline[0]:
            o.time_verify_core(3, A, z, c, t1, h)
            n += 16
        done[i] = n

    ths = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter()
    deadline[0] = t0 + target_s
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    quota, qsrc = cpu_quota()
    return {"value": sum(done) / dt, "unit": "verify/s", "threads": nthreads, "cores": round(quota, 2) if quota else nthreads,
            "kind": "port", "sample": f"{sum(done)} level-3 verify cores on {nthreads} threads in {dt:.1f} s; CPU quota: "
                                      f"{'%.2f CPUs' % quota if quota else 'none'} ({qsrc})"}


def _cpu_threads():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def _cpu_leg(make_work, units_per_call, unit, what, target_s=3.0, target_all_s=3.0):
    """One thread, then every visible hardware thread, of an oracle batch call (ctypes drops the GIL in the foreign call): make_work(i)
    returns thread i's closure over its own inputs; time-bounded, quota-aware like the NTT leg.  kind 'port': the reference has no
    software form of the fused op-graphs (they are RTL: combined_top.v), the oracle's C restatement is what a CPU runs."""
    import threading
    w0 = make_work(0)
    w0()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < target_s:
        w0()
        n += units_per_call
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": unit, "cores": 1, "kind": "port", "sample": f"{n} {what} (oracle C restatement) in {dt:.1f} s, 1 thread"}
    nthreads = _cpu_threads()
    works = [make_work(i % 8) for i in range(nthreads)]
    done = [0] * nthreads
    deadline = [0.0]

    def run(i):
        k = 0
        while time.perf_counter() < deadline[0]:
            works[i]()
            k += units_per_call
        done[i] = k
    ths = [threading.Thread(target=run, args=(i,)) for i in range(nthreads)]
    t0 = time.perf_counter()
    deadline[0] = t0 + target_all_s
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    dt = time.perf_counter() - t0
    quota, qsrc = cpu_quota()
    out["all_threads"] = {"value": sum(done) / dt, "unit": unit, "threads": nthreads, "cores": round(quota, 2) if quota else nthreads, "kind": "port",
                          "sample": f"{sum(done)} {what} on {nthreads} threads in {dt:.1f} s; CPU quota: "
                                    f"{'%.2f CPUs' % quota if quota else 'none'} ({qsrc})"}
    return out


def cpu_baseline_matvec(target_s=3.0):
    """BASELINE configs[2] on the host: level-2 (K = L = 4) A.y, a matrix per item (combined_top.v:1850-1933), oracle orc_matvec"""
    from oracle.oracle import Oracle, splitmix64_polys
    o = Oracle()
    n = 64

    def make(i):
        A = splitmix64_polys(n * 16, seed=500 + i).reshape(n, 4, 4, 256)
        y = splitmix64_polys(n * 4, seed=600 + i).reshape(n, 4, 256)
        return lambda: o.matvec(4, 4, A, y)
    return _cpu_leg(make, n, "matvec/s", "level-2 mat-vecs, a matrix per item", target_s, target_s)


def cpu_baseline_sign_attempt(target_s=3.0):
    """BASELINE configs[4]'s unit on the host: one level-5 sign attempt = phase 1 (w = A.y, decompose) + phase 2 (z, r0 / hint checks) under
    one key (combined_top.v:1830-2229), oracle orc_sign_phase1 + orc_sign_phase2"""
    from oracle.oracle import Oracle, splitmix64_polys, Q
    o = Oracle()
    n, K, L = 16, 8, 7

    def make(i):
        rng = np.random.default_rng(700 + i)
        A = splitmix64_polys(K * L, seed=800 + i).reshape(1, K, L, 256)
        y = np.mod(rng.integers(-(1 << 19) + 1, (1 << 19) + 1, (n, L, 256)), Q).astype(np.int32)
        c = np.zeros((n, 256), np.int32)
        c[:, ::5] = 1
        c[:, 1::7] = Q - 1
        s1, s2, t0 = (splitmix64_polys(m, seed=900 + i + j).reshape(1, m, 256) for j, m in enumerate((L, K, K)))

        def work():
            w1, w0 = o.sign_phase1(5, A, y)
            o.sign_phase2(5, c, y, w0, w1, s1, s2, t0)
        return work
    return _cpu_leg(make, n, "attempt/s", "level-5 sign attempts (phase 1 + phase 2), one key", target_s, target_s)


def synth_verify(n, seed):
    from oracle.oracle import splitmix64_polys, Q, N
    K, L, tau, g1 = 6, 5, 49, 1 << 19
    rng = np.random.default_rng(seed)
    A = splitmix64_polys(n * K * L, seed=seed).reshape(n, K, L, N)
    z = np.mod(rng.integers(-(g1 - 1), g1 + 1, (n, L, N)), Q).astype(np.int32)
    c = np.zeros((n, N), np.int32)
    cols = np.argsort(rng.random((n, N)), axis=1)[:, :tau]
    sg = np.where(rng.integers(0, 2, (n, tau)) == 1, 1, Q - 1).astype(np.int32)
    np.put_along_axis(c, cols, sg, axis=1)
    t1 = rng.integers(0, 1 << 10, (n, K, N)).astype(np.int32)
    h = (rng.random((n, K, N)) < 0.03).astype(np.uint8)
    return A, z, c, t1, h


def system_clocks():
    """best-effort sclk / mclk [MHz] from the driver (amdsmi, then sysfs, then rocm-smi); {} where the container shows none"""
    out = {}
    try:
        import glob
        for card in sorted(glob.glob("/sys/class/drm/card*/device")):
            for name, key in (("pp_dpm_sclk", "sclk_mhz"), ("pp_dpm_mclk", "mclk_mhz")):
                p = os.path.join(card, name)
                if key not in out and os.path.exists(p):
                    cur = [ln for ln in open(p).read().splitlines() if ln.strip().endswith("*")]
                    if cur:
                        out[key] = float(cur[0].split(":")[1].strip().split("M")[0])
                        out["source"] = os.path.dirname(p) + "/pp_dpm_{sclk,mclk}, read AFTER the timed regions (the GPU is idle by then: " \
                                        "the clock under load is shader_mhz_*)"
            if out:
                return out
    except Exception:
        pass
    try:
        import subprocess
        txt = subprocess.run(["rocm-smi", "-c", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(txt)
        card = d[sorted(d)[0]]
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl and "sclk_mhz" not in out:
                out["sclk_mhz"] = float("".join(ch for ch in v if ch.isdigit() or ch == "."))
            if "mclk" in kl and "mclk_mhz" not in out:
                out["mclk_mhz"] = float("".join(ch for ch in v if ch.isdigit() or ch == "."))
        if out:
            out["source"] = "rocm-smi -c --json"
    except Exception:
        pass
    return out


def _pmc_summary(family):
    """profiles/pmc_summary.json if its stamps (git blob ids of the sources the kernel family is compiled from, written by
    scripts/pmc_summary.py) still match the tree; None otherwise -- a committed counter figure must not outlive the kernel it was measured on"""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        d = json.load(open(p))
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from pmc_summary import source_stamps
        return d if d.get("_source_blobs", {}).get(family) == source_stamps()[family] else None
    except Exception:
        return None


def pmc_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/), if there is one measured on these kernel sources"""
    d = _pmc_summary("ntt" if kernel_key.startswith("ntt") else "verify")
    return d.get(kernel_key, {}).get("hbm_bytes_per_launch") if d else None


def pmc_sign_valu():
    """VALU instructions per level-5 sign attempt (phase 1, phase 2) from the committed SQ_INSTS_VALU passes, if any (and not stale)"""
    d = _pmc_summary("sign")
    try:
        v = d.get("sign_valu_insts_per_attempt") if d else None
        return {"phase1": float(v["phase1"]), "phase2": float(v["phase2"])} if v else None
    except Exception:
        return None


class Bench:
    """what the legs share: arguments, the library, streams, the timing helpers"""

    def __init__(self, args):
        from dilithium_amd import api, sharding
        from dilithium_amd import lib as dlib
        import ctypes as C
        self.args, self.api, self.sharding, self.dlib, self.C = args, api, sharding, dlib, C
        self.rank, self.world, self.local = sharding.init_distributed()
        assert self.world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={self.world}"
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        self.dev = sharding.local_device(self.local)
        torch.cuda.set_device(self.dev)
        api.init(self.dev)
        self.L = dlib.load()
        self.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        self.probe_stream = torch.cuda.Stream()
        self.probe_buf = torch.zeros(4, dtype=torch.int64, device="cuda")
        self.probe_effect = {}
        self.last_regions = []
        self.NS = max(1, args.streams)
        self.tstreams = [torch.cuda.Stream() for _ in range(self.NS)] if self.NS > 1 else [torch.cuda.current_stream()]
        self.hstreams = [C.c_void_p(ts.cuda_stream) for ts in self.tstreams]

    def ev(self):
        e = self.C.c_void_p()
        self.dlib.check(self.L.dil_event_create(self.C.byref(e)))
        return e

    def elapsed(self, a, b):
        ms = self.C.c_float()
        self.dlib.check(self.L.dil_event_elapsed_ms(self.C.byref(ms), a, b))
        return float(ms.value)

    def timed(self, fn, st=None, min_ms=None):
        """average milliseconds per call of fn() (direct C-ABI launches on stream `st`): HIP events around a region that is
        at least min_ms long -- the repeat count comes from a calibration pass, not from --steps.  One region unmeasured, then three measured ones of which the
        MEDIAN counts: the composite calls synchronise with the host every rejection round, and one descheduling of this
        process inside a 25-ms region was seen to halve a rate (profiles/r04zz_bench.log, sign at 8192: 3.9 M/s beside 6.5-6.8
        in the same visit's other two bench runs); the median shrugs one such region off without favouring a fast one (the
        fastest of the three can be a region at a higher clock).  Returns (ms per call, calls in a measured region)."""
        L, dlib = self.L, self.dlib
        st = self.stream if st is None else st
        min_ms = self.args.min_ms if min_ms is None else min_ms
        e0, e1 = self.ev(), self.ev()
        reps, used, rc, pers = 4, 4, 0, []
        for phase in range(6):                   # warm-up, calibrate, one full region unmeasured, then the three measured regions
            L.dil_event_record(e0, st)           # (the chip needs > 50 ms of a kernel's load to settle: the fused verify core ran
            for i in range(reps):                #  63.1 us per launch in the region after the calibration pass and 59.0 / 59.1 in
                rc |= fn(i)                      #  the next two -- profiles/r04w_verify_transient.txt has the curve after idle)
            L.dil_event_record(e1, st)
            used = reps
            per = max(self.elapsed(e0, e1) / reps, 1e-4)
            if phase >= 3:
                pers.append(per)
            elif phase < 2:
                reps = max(10, int(min_ms / per) + 1)
        dlib.check(rc, "timed launches")
        self.last_regions = list(pers)
        return sorted(pers)[1], used

    def with_clock(self, fn, span_ms, tag=None):
        """fn() ALONE on the device is the measurement; then fn() once more while a one-lane probe kernel on a side stream
        measures the effective shader clock over about span_ms (csrc/kernels.hip clock_probe_kernel: shader cycles per 100 MHz
        tick).  Two passes because the probe is not free for every kernel: a second active queue was seen to cost the fused
        verify kernel 5-6 % (63.4 vs 59.5 us per launch), which is instrumentation, not the kernel.  Returns (the first pass's
        result, MHz or None); probe_effect[tag] keeps the second pass's result beside it."""
        res = fn()
        try:
            torch.cuda.synchronize()
            self.dlib.check(self.L.dil_clock_probe_dev(self.P(self.probe_buf), max(1000, int(span_ms * 1000)),
                                                        self.C.c_void_p(self.probe_stream.cuda_stream)), "clock probe")
            again = fn()
            self.probe_stream.synchronize()
            if tag:
                self.probe_effect[tag] = again
            c0, c1, r0, r1 = [int(x) for x in self.probe_buf.cpu().tolist()]
            mhz = (c1 - c0) / max(1, r1 - r0) * 100.0
            return res, (mhz if 200.0 < mhz < 5000.0 else None)
        except Exception:   # noqa: BLE001
            return res, None


def leg_headline(cx):
    """BASELINE configs[1]: the timed regions of `value`, the one-stream per-kernel roofline, its in-run yardstick (the traffic-only
    skeleton), the LLC-resident variant.  Returns the JSON line's top level."""
    args, api, sharding, dlib, C, L, P, stream = cx.args, cx.api, cx.sharding, cx.dlib, cx.C, cx.L, cx.P, cx.stream
    rank, world, NS, hstreams, timed, with_clock, probe_effect = cx.rank, cx.world, cx.NS, cx.hstreams, cx.timed, cx.with_clock, cx.probe_effect
    # ---- inputs, resident in HBM ------------------------------------------------------------
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    R = max(1, args.rotate)
    R += (-R) % NS                                        # a batch must always meet the same stream
    bufs = [torch.randint(0, 8380417, (BATCH, 256), dtype=torch.int32, device="cuda", generator=g) for _ in range(R)]
    check = bufs[0][:64].clone()
    torch.cuda.synchronize()

    ptrs = [P(b) for b in bufs]                           # direct C-ABI calls: minimal host overhead
    def step(i, fixed=None, one=False):
        p = ptrs[(i % R) if fixed is None else (fixed + i % NS)]
        st = hstreams[0] if one else hstreams[i % NS]
        return L.dil_ntt_dev(p, BATCH, st) | L.dil_invntt_dev(p, BATCH, st)

    def region(k, fixed=None, one=False):
        """EXACTLY k steps between barrier + synchronize on both sides, host clock; no events inside (an event record between two
        launches costs several microseconds of GPU idle time -- at 26 us per kernel a 10 % perturbation).  Returns this rank's
        wall seconds."""
        sharding.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = 0
        for i in range(k):
            rc |= step(i, fixed, one)
        torch.cuda.synchronize()
        sharding.barrier()
        wall = time.perf_counter() - t0
        dlib.check(rc, "timed NTT launches")
        return wall

    def measure(k, fixed=None, one=False, min_region_ms=None, min_total_ms=None, min_regions=None):
        """Robust at any --steps: a timed region is M back-to-back groups of K steps (M = what makes it >= --region-ms long, so the
        ~30 us a synchronised region loses to the first launch's ramp and the last one's drain stays below 0.1 %), and R >= --regions
        such regions are timed (>= --total-ms in all).  Returns per-step seconds: median / min / max over the regions (each the MAX
        over ranks), M, R."""
        min_region_ms = args.region_ms if min_region_ms is None else min_region_ms
        min_total_ms = args.total_ms if min_total_ms is None else min_total_ms
        min_regions = args.regions if min_regions is None else min_regions
        probe = max(region(k, fixed, one), 1e-6)                                         # calibration (also a warm region)
        m = max(1, int(np.ceil(min_region_ms * 1e-3 / probe)))
        m = int(sharding.max_over_ranks(float(m)))                                       # same M on every rank
        r = max(min_regions, int(np.ceil(min_total_ms / max(min_region_ms, probe * m * 1e3))))
        r = int(sharding.max_over_ranks(float(r)))                                       # a region holds barriers: same R on every rank too
        walls = torch.tensor([region(m * k, fixed, one) for _ in range(r)], dtype=torch.float64)
        if world > 1:
            w = walls.cuda()
            torch.distributed.all_reduce(w, op=torch.distributed.ReduceOp.MAX)
            walls = w.cpu()
        per_step = (walls / (m * k)).numpy()
        return {"median": float(np.median(per_step)), "min": float(per_step.min()), "max": float(per_step.max()), "groups_per_region": m,
                "regions": r}

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:      # untimed clock/power warm-up
        for i in range(32):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    K = args.steps
    hd = measure(K)
    assert torch.equal(bufs[0][:64], check), "fwd+inv round trip is not the identity"
    step_s = hd["median"]
    value = world * 2 * BATCH / step_s
    overlapped_gbs = NTT_BYTES * BATCH * 2 / step_s / 1e9          # the SAME clock as `value`: all launches' bytes / elapsed time

    # the same steps on ONE stream: the per-kernel roofline (no overlap between launches; each launch's share of the region
    # includes its ~2 us dispatch gap, so this is a lower bound of what rocprofv3 reports per kernel)
    one, ntt_mhz = with_clock(lambda: measure(K, one=True, min_total_ms=args.total_ms / 2, min_regions=max(3, args.regions // 2)),
                              args.total_ms / 2)
    one_stream_value = world * 2 * BATCH / one["median"]
    launch_ms = one["median"] / 2 * 1e3
    kernel_gbs = NTT_BYTES * BATCH / (launch_ms * 1e-3) / 1e9

    # what the transforms' access pattern reaches on THIS box without any arithmetic: the same launch shape, loads and stores
    # (csrc/kernels.hip ntt_traffic_kernel), on scratch copies of the rotating batches, one stream, same clock
    scratch = [torch.randint(0, 8380417, (BATCH, 256), dtype=torch.int32, device="cuda", generator=g) for _ in range(R)]
    sptr = [P(b) for b in scratch]

    def traffic_region(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = 0
        for i in range(k):
            rc |= L.dil_ntt_traffic_dev(sptr[i % R], BATCH, 0, hstreams[0]) | L.dil_ntt_traffic_dev(sptr[i % R], BATCH, 1, hstreams[0])
        torch.cuda.synchronize()
        dlib.check(rc, "traffic-only launches")
        return (time.perf_counter() - t0) / (2 * k)
    traffic_region(8)
    kk = max(8, int(0.05 / max(traffic_region(8), 1e-6) / 2))
    ach_s = float(np.median([traffic_region(kk) for _ in range(5)]))
    achievable_gbs = NTT_BYTES * BATCH / ach_s / 1e9
    del scratch

    # LLC-resident variant (the same NS batches every step: 64 MiB each, inside the 256 MiB Infinity Cache) for context
    for i in range(8):
        step(i, fixed=0)
    torch.cuda.synchronize()
    llc = measure(K, fixed=0, min_total_ms=args.total_ms / 4, min_regions=3)
    llc_value = world * 2 * BATCH / llc["median"]

    traffic = pmc_traffic("ntt_fwd_kernel")
    out = {
        "metric": "ntt256_transforms_per_sec", "value": value, "unit": "NTT/s", "n_gpus": world, "steps": K,
        "warmup": args.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: batched forward+inverse NTT, n=256, q=8380417, "
                               "batch=65536 polynomials per GPU; step = 1 fwd launch + 1 inv launch",
                   "batch_per_gpu": BATCH, "rotating_resident_batches": R, "streams": NS,
                   "parallelism": f"shard x{world}", "bytes_per_transform": NTT_BYTES},
        "timing": {"clock": "host perf_counter around barrier + torch.cuda.synchronize() on both sides, MAX over ranks per region",
                   "steps_per_timed_region": hd["groups_per_region"] * K, "regions": hd["regions"],
                   "ms_per_step_median": hd["median"] * 1e3, "ms_per_step_min": hd["min"] * 1e3, "ms_per_step_max": hd["max"] * 1e3,
                   "value_min": world * 2 * BATCH / hd["max"], "value_max": world * 2 * BATCH / hd["min"],
                   "note": "a timed region is a whole number of K-step groups, long enough (--region-ms) that the result does not "
                           "depend on --steps; value / ms_per_step are the MEDIAN region"},
        "roofline": {"bound": "hbm",
                     "kernel": "ntt_fwd_kernel<LAYOUT_POLY> / ntt_inv_kernel<LAYOUT_POLY> (alternating launches, identical "
                               "algorithmic bytes)",
                     "achieved": kernel_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kernel_gbs / HBM_PEAK_GBS,
                     "avg_launch_ms": launch_ms, "algorithmic_bytes_per_launch": NTT_BYTES * BATCH,
                     "achieved_overlapped": overlapped_gbs, "frac_overlapped": overlapped_gbs / HBM_PEAK_GBS,
                     "concurrent_launches_overlapped": NS,
                     "achievable": achievable_gbs, "achievable_frac_of_peak": achievable_gbs / HBM_PEAK_GBS,
                     "frac_of_achievable": kernel_gbs / achievable_gbs,
                     "achievable_source": "measured in this run: ntt_traffic_kernel = the transforms' loads and stores (same grid, same "
                                          "prefetch, strided 256-B dword accesses on one side and 1-KiB dwordx4 rows on the other) with NO "
                                          "arithmetic, forward + inverse pattern alternating on one stream over the same rotating batches",
                     "traffic": traffic,
                     "traffic_level": "L2 <-> memory-fabric bytes (TCC FETCH_SIZE x 2 x 1024 B + WRITE_SIZE x 1024 B, the gfx950 correction of "
                                      "MI355X_MICROARCH.md); the counter cannot tell Infinity-Cache hits from HBM accesses -- the steps rotate over "
                                      "512 MiB of batches (twice the 256 MiB Infinity Cache), which makes them HBM reads for this kernel",
                     "traffic_source": "profiles/pmc_summary.json: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                       "this kernel (bytes per launch; not re-measured in this run)" if traffic else None,
                     "timing": "every figure on the host clock of `value` (barrier + synchronize on both sides of a region): achieved / frac "
                               "= algorithmic bytes per launch / avg_launch_ms, the per-launch share of regions on ONE stream (median of "
                               f"{one['regions']} regions); achieved_overlapped / frac_overlapped = bytes per step / ms_per_step of `value`, "
                               "with `concurrent_launches_overlapped` launches in flight"},
        "llc_resident_value": llc_value,
        "one_stream_value": one_stream_value,
        "clocks": dict(system_clocks(), shader_mhz_during_one_stream_ntt=ntt_mhz,
                       shader_mhz_source="in-kernel probe on a side stream over the one-stream regions: shader cycle counter (s_memtime) "
                                         "against the constant 100 MHz counter (s_memrealtime); data-sheet maximum 2400"),
    }
    return out


def leg_distributed(cx, out):
    """falsifiable on the multi-GPU node: what RCCL itself says about the job, and where every rank sits"""
    rank, world = cx.rank, cx.world
    ones = torch.ones(1, device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu")
    torch.distributed.all_reduce(ones)
    props = torch.cuda.get_device_properties(cx.dev)
    mine = {"rank": rank, "local_rank": cx.local, "device": int(cx.dev), "name": props.name,
            "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", ""))}
    everyone = [None] * world
    torch.distributed.all_gather_object(everyone, mine)
    out["distributed"] = {"backend": torch.distributed.get_backend(), "rccl_nranks": int(round(float(ones.item()))),
                          "world_size": torch.distributed.get_world_size(), "ranks": everyone,
                          "distinct_devices": len({(e["device"], e["pci_bus_id"], e["uuid"]) for e in everyone})}
    try:      # which RCCL this is (multi_gpu.hip binds the same library at run time)
        out["distributed"]["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
        import ctypes.util
        out["distributed"]["librccl"] = next((ln.split()[-1] for ln in open("/proc/self/maps") if "librccl" in ln), ctypes.util.find_library("rccl"))
    except Exception as e:  # noqa: BLE001
        out["distributed"]["rccl_version"] = repr(e)


def leg_verify_core(cx):
    """BASELINE configs[3]: the fused level-3 verify core, a key per item, rotating over input sets larger than the Infinity Cache"""
    args, api, sharding, dlib, C, L, P, stream = cx.args, cx.api, cx.sharding, cx.dlib, cx.C, cx.L, cx.P, cx.stream
    rank, world, NS, hstreams, timed, with_clock, probe_effect = cx.rank, cx.world, cx.NS, cx.hstreams, cx.timed, cx.with_clock, cx.probe_effect
    cu = lambda x: torch.from_numpy(x).cuda()  # noqa: E731
    # rotate over FOUR input sets (1.44 GB, 5.6 x the 256 MiB Infinity Cache).  Rounds 1-4 rotated over two (720 MiB) and called that
    # HBM-streaming; it is not: the cache keeps part of a set across one intervening launch -- 61.7 us per launch over two sets, 71.4 over
    # three, four, six or eight (profiles/r05i_rotating_sets.txt).  The two-set figure stays beside this one as `two_rotating_sets`.
    VSETS = 4
    vsets = []
    for j in range(VSETS):
        A, z, c, t1_, h = synth_verify(VBATCH, 77 + rank + 100 * j)
        if os.environ.get("DIL_BENCH_VERIFY_UNIFORM"):     # experiment: uniform residues everywhere, as scripts/ab_verify.py feeds the kernel
            gz = torch.Generator(device="cuda").manual_seed(j)
            ur = lambda *sh: torch.randint(0, 8380417, sh, dtype=torch.int32, device="cuda", generator=gz)  # noqa: E731
            vsets.append((ur(VBATCH, 6, 5, 256), ur(VBATCH, 5, 256), ur(VBATCH, 256), cu(t1_), cu(h),
                          torch.empty((VBATCH, 6, 256), dtype=torch.uint8, device="cuda")))
            continue
        vsets.append((cu(A), cu(z), cu(c), cu(t1_), cu(h), torch.empty((VBATCH, 6, 256), dtype=torch.uint8, device="cuda")))
    dA, dz, dc, dt1, dh, w1 = vsets[0]
    vptr = [[P(t) for t in vs_] for vs_ in vsets]
    torch.cuda.synchronize()

    def vstep(i):
        pA, pz, pc, pt1, ph, pw1 = vptr[i % VSETS]
        return L.dil_verify_core_dev(pw1, pA, pz, pc, pt1, ph, 3, VBATCH, 0, stream)

    sharding.barrier()
    v_regions = []

    def timed_v():
        r = timed(vstep)
        v_regions.append(list(cx.last_regions))
        return r
    (v_ms, v_reps), v_mhz = with_clock(timed_v, 5 * args.min_ms, "verify")
    v_ms = sharding.max_over_ranks(v_ms)
    v_gbs = VERIFY3_BYTES * VBATCH / (v_ms * 1e-3) / 1e9
    vtraffic = pmc_traffic("verify_kernel")
    sec = {"metric": "dilithium3_verify_cores_per_sec", "value": world * VBATCH / (v_ms * 1e-3), "unit": "verify/s",
           "config": {"workload": "BASELINE configs[3]: level-3 verify core (NTT z, A.z - c.t1.2^d, INTT, "
                                  "UseHint -> w1), batch=8192 per GPU, distinct pk (A, t1 per item)",
                      "bytes_per_verify": VERIFY3_BYTES, "streams": 1, "rotating_input_sets": VSETS, "timed_launches": v_reps},
           "roofline": {"bound": "hbm", "kernel": "verify_wpi_kernel<3>", "achieved": v_gbs, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": v_gbs / HBM_PEAK_GBS, "traffic": vtraffic,
                        "traffic_source": "profiles/pmc_summary.json (committed PMC passes)" if vtraffic else None,
                        "avg_launch_ms": v_ms, "shader_mhz_observed": v_mhz,
                        "avg_launch_ms_beside_clock_probe": probe_effect.get("verify", (None,))[0],
                        "region_ms": {"alone": v_regions[0] if v_regions else None,
                                      "beside_clock_probe": v_regions[1] if len(v_regions) > 1 else None}}}
    # the same launches alternating over TWO streams (input set j on stream j): the next launch's ramp hides the
    # previous one's tail and the dispatch gap -- the aggregate rate, reported beside the single-kernel fraction
    if NS > 1 and not args.no_verify_overlap:
        def vstep2(i):
            pA, pz, pc, pt1, ph, pw1 = vptr[i % VSETS]
            return L.dil_verify_core_dev(pw1, pA, pz, pc, pt1, ph, 3, VBATCH, 0, hstreams[i % 2])     # (set i % 4 always meets stream i % 2)
        reps2 = max(20, 2 * (v_reps // 2))
        rc2 = 0
        for i in range(8):
            rc2 |= vstep2(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(reps2):
            rc2 |= vstep2(i)
        torch.cuda.synchronize()
        dlib.check(rc2, "overlapped verify launches")
        v2_ms = sharding.max_over_ranks((time.perf_counter() - t0) / reps2 * 1e3)
        v2_gbs = VERIFY3_BYTES * VBATCH / (v2_ms * 1e-3) / 1e9
        sec["roofline"].update({"achieved_overlapped": v2_gbs, "frac_overlapped": v2_gbs / HBM_PEAK_GBS,
                                "concurrent_launches_overlapped": 2, "overlapped_value": world * VBATCH / (v2_ms * 1e-3),
                                "timing": "frac = per-launch share of back-to-back launches on ONE stream (HIP events, the median of three regions); "
                                          "frac_overlapped = the same launches alternating over two streams, host clock around "
                                          "a synchronised region"})
    # the same launches over ONE input set (360 MiB, partly served by the 256 MiB Infinity Cache), for context
    l_ms, _ = timed(lambda i: L.dil_verify_core_dev(vptr[0][5], vptr[0][0], vptr[0][1], vptr[0][2], vptr[0][3], vptr[0][4], 3,
                                                    VBATCH, 0, stream))
    sec["llc_assisted_value"] = world * VBATCH / (l_ms * 1e-3)
    # same pipeline with ONE public key for the whole batch (A, t1 staged in LDS): VALU-bound, reported beside it
    s_ms, _ = timed(lambda i: L.dil_verify_core_dev(vptr[i % VSETS][5], vptr[0][0], vptr[i % VSETS][1], vptr[i % VSETS][2],
                                                    vptr[0][3], vptr[i % VSETS][4], 3, VBATCH, 1, stream))
    sec["shared_pk"] = {"value": VBATCH / (s_ms * 1e-3), "unit": "verify/s per GPU", "avg_launch_ms": s_ms,
                        "bytes_per_verify": 15 * 1024, "kernel": "verify_shared_kernel<3,16>",
                        "bound": "valu (key material LDS-resident)"}
    # the same over TWO rotating sets, as rounds 1-4 measured this leg: partly served by the Infinity Cache
    t_ms, _ = timed(lambda i: L.dil_verify_core_dev(vptr[i % 2][5], vptr[i % 2][0], vptr[i % 2][1], vptr[i % 2][2], vptr[i % 2][3], vptr[i % 2][4], 3,
                                                    VBATCH, 0, stream))
    sec["two_rotating_sets"] = {"value": world * VBATCH / (t_ms * 1e-3), "avg_launch_ms": t_ms, "bytes_resident": 2 * VERIFY3_BYTES * VBATCH,
                                "frac": VERIFY3_BYTES * VBATCH / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "note": "LLC-assisted: what BENCH_r01..r04 reported as the secondary value"}
    return sec


def leg_other_configs(cx, sec):
    """configs[2] (level-2 mat-vec, per-item and shared matrix) and configs[4]'s per-GPU slice (level-5 sign attempt), for the record"""
    args, api, sharding, dlib, C, L, P, stream = cx.args, cx.api, cx.sharding, cx.dlib, cx.C, cx.L, cx.P, cx.stream
    rank, world, NS, hstreams, timed, with_clock, probe_effect = cx.rank, cx.world, cx.NS, cx.hstreams, cx.timed, cx.with_clock, cx.probe_effect
    # configs[2] and configs[4] of BASELINE.json (parity-test configs; timed here for the record only)
    g2 = torch.Generator(device="cuda").manual_seed(5)
    rnd = lambda *sh: torch.randint(0, 8380417, sh, dtype=torch.int32, device="cuda", generator=g2)  # noqa: E731
    try:
        A2 = [rnd(4096, 4, 4, 256) for _ in range(16)]          # 16 x 64 MiB of A (4 x the Infinity Cache): rotating, HBM-streaming (rounds 1-4: 4 matrices = 256 MiB, LLC-assisted)
        y2, wout = rnd(4096, 4, 256), torch.empty((4096, 4, 256), dtype=torch.int32, device="cuda")
        m_ms, _ = timed(lambda i: L.dil_matvec_dev(P(wout), P(A2[i % 16]), P(y2), 2, 4096, 0, stream))
        ms_ms, _ = timed(lambda i: L.dil_matvec_dev(P(wout), P(A2[0]), P(y2), 2, 4096, 1, stream))     # ONE matrix for the batch (report both)
        # a real key and real challenges (phase 2 reads c s1 and c s2 off one transform, exact for valid inputs: DESIGN.md 4)
        small = lambda lim, *sh: (torch.randint(-lim, lim + 1, sh, dtype=torch.int64, device="cuda", generator=g2) % 8380417).to(torch.int32)  # noqa: E731
        A5, y5 = rnd(1, 8, 7, 256), small((1 << 19) - 1, 8192, 7, 256)
        c5 = api.sample_in_ball(torch.randint(0, 256, (8192, 32), dtype=torch.uint8, device="cuda", generator=g2), 5)
        s1h, s2h, t0h = small(2, 1, 7, 256), small(2, 1, 8, 256), small(4095, 1, 8, 256)
        for t in (s1h, s2h, t0h):
            api.ntt(t)
        w1s = torch.empty((8192, 8, 256), dtype=torch.uint8, device="cuda")
        w0s = torch.empty((8192, 8, 256), dtype=torch.int32, device="cuda")
        z5 = torch.empty((8192, 7, 256), dtype=torch.int32, device="cuda")
        h5 = torch.empty((8192, 8, 256), dtype=torch.uint8, device="cuda")
        f5 = torch.empty((8192,), dtype=torch.int32, device="cuda")

        def attempt(i):
            return L.dil_sign_phase1_dev(P(w1s), P(w0s), P(A5), P(y5), 5, 8192, 1, stream) | \
                L.dil_sign_phase2_skey_dev(P(z5), P(h5), P(f5), P(c5), P(y5), P(w0s), P(w1s), P(s1h), P(s2h), P(t0h), 5, 8192, 1, 0, stream)
        (a_ms, _), sign_mhz = with_clock(lambda: timed(attempt), 5 * args.min_ms, "attempt")
        # the same pair over TWO rotating sets of buffers (2 x 223 MB: past the 256 MiB Infinity Cache), as the headline and the
        # verify leg are measured: the HBM-streaming figure.  (One set = 223 MB is largely cache-resident.)
        y5b = small((1 << 19) - 1, 8192, 7, 256)
        c5b = api.sample_in_ball(torch.randint(0, 256, (8192, 32), dtype=torch.uint8, device="cuda", generator=g2), 5)
        rot = [(y5, c5, w1s, w0s, z5, h5), (y5b, c5b, torch.empty_like(w1s), torch.empty_like(w0s), torch.empty_like(z5), torch.empty_like(h5))]

        def attempt_rot(i):
            yy, cc, ww1, ww0, zz, hh = rot[i & 1]
            return L.dil_sign_phase1_dev(P(ww1), P(ww0), P(A5), P(yy), 5, 8192, 1, stream) | \
                L.dil_sign_phase2_skey_dev(P(zz), P(hh), P(f5), P(cc), P(yy), P(ww0), P(ww1), P(s1h), P(s2h), P(t0h), 5, 8192, 1, 0, stream)
        ar_ms, _ = timed(attempt_rot)
        p1_ms, _ = timed(lambda i: L.dil_sign_phase1_dev(P(w1s), P(w0s), P(A5), P(y5), 5, 8192, 1, stream))
        p2_ms, _ = timed(lambda i: L.dil_sign_phase2_skey_dev(P(z5), P(h5), P(f5), P(c5), P(y5), P(w0s), P(w1s), P(s1h), P(s2h), P(t0h), 5, 8192, 1,
                                                              0, stream))
        sv = pmc_sign_valu()
        sec["other_configs"] = {
            "configs[2] level-2 A.y matvec batch=4096 distinct A (16 rotating matrices)": {
                "matvecs_per_s": 4096 / (m_ms * 1e-3), "ms": m_ms, "GBps": 24 * 1024 * 4096 / (m_ms * 1e-3) / 1e9,
                "bytes_per_matvec": 24 * 1024, "frac_of_hbm_peak": 24 * 1024 * 4096 / (m_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernel": "matvec_wpi_kernel<4,4,2,OUT_W>"},
            "configs[2] level-2 A.y matvec batch=4096 shared A (one matrix, LDS-resident)": {
                "matvecs_per_s": 4096 / (ms_ms * 1e-3), "ms": ms_ms, "bytes_per_matvec": 8 * 1024,
                "GBps": (8 * 1024 * 4096 + 16 * 1024) / (ms_ms * 1e-3) / 1e9,
                "frac_of_hbm_peak": (8 * 1024 * 4096 + 16 * 1024) / (ms_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernel": "matvec_shared_kernel<4,4,2,OUT_W,16>", "bound": "valu / launch (32 MiB of traffic)"},
            "configs[4] level-5 sign attempt (phase1+phase2) batch=8192 per GPU, shared key": {
                "attempts_per_s": 8192 / (a_ms * 1e-3), "ms": a_ms, "phase1_ms": p1_ms, "phase2_ms": p2_ms,
                "buffers": "one set (y, c, w1, w0, z, h = 223 MB): largely Infinity-Cache-resident, as in rounds 1-3",
                "hbm_streaming": {"attempts_per_s": 8192 / (ar_ms * 1e-3), "ms": ar_ms, "buffers": "two rotating sets (446 MB)",
                                  "frac_of_hbm_peak": (46 * 1024) * 8192 / (ar_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "roofline": None if not sv else {
                    "bound": "valu", "unit": "wave64 VALU instructions/s",
                    "peak": VALU_PEAK, "peak_source": "256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction (MI355X_MICROARCH.md)",
                    "valu_insts_per_attempt": sv, "source": "profiles/pmc_summary.json (committed SQ_INSTS_VALU passes of these kernels / 8192)",
                    "phase1_frac": sv["phase1"] * 8192 / (p1_ms * 1e-3) / VALU_PEAK,
                    "phase2_frac": sv["phase2"] * 8192 / (p2_ms * 1e-3) / VALU_PEAK,
                    "shader_mhz_observed": sign_mhz, "attempt_ms_beside_clock_probe": probe_effect.get("attempt", (None,))[0],
                    "peak_at_observed_clock": None if not sign_mhz else VALU_PEAK * sign_mhz / 2400.0,
                    "phase1_frac_at_observed_clock": None if not sign_mhz else sv["phase1"] * 8192 / (p1_ms * 1e-3) / (VALU_PEAK * sign_mhz / 2400.0),
                    "phase2_frac_at_observed_clock": None if not sign_mhz else sv["phase2"] * 8192 / (p2_ms * 1e-3) / (VALU_PEAK * sign_mhz / 2400.0),
                    "hbm": {"bytes_per_attempt": 45 * 1024 + 1024, "note": "y 7 + w0 8 + w1 2 + w1 packed 1 KiB (phase 1) and c 1 + y 7 + "
                            "w0 8 + w1 2 + z 7 + h 2 KiB (phase 2): the int32 planes of the public entry points",
                            "frac_of_hbm_peak": (46 * 1024) * 8192 / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                    "note": "the measured issue cost of this instruction mix is ~4.3 cycles (multiplies 4.4, adds 2.5), and the "
                            "transform alone reaches 0.62-0.67 of this peak in a compute-only loop (profiles/r03c_tune_xchg.txt)"}}}
    except Exception as e:  # noqa: BLE001
        sec["other_configs"] = {"error": repr(e)}
    try:      # the reference's polymul chain (ntt, ntt, pointwise_barrett, invntt; ntt2x2_test.cpp:109-137): fused kernel vs the four launches
        gp = torch.Generator(device="cuda").manual_seed(11)
        NP = 32768
        pa = [torch.randint(0, 8380417, (NP, 256), dtype=torch.int32, device="cuda", generator=gp) for _ in range(8)]      # 8 x (a, b, c) x 32 MiB = 768 MiB rotating
        pb = [torch.randint(0, 8380417, (NP, 256), dtype=torch.int32, device="cuda", generator=gp) for _ in range(8)]
        pc = [torch.empty((NP, 256), dtype=torch.int32, device="cuda") for _ in range(8)]
        f_ms, _ = timed(lambda i: L.dil_polymul_dev(P(pc[i % 8]), P(pa[i % 8]), P(pb[i % 8]), NP, stream))

        def chain(i):
            a_, b_ = P(pa[i % 8]), P(pb[i % 8])
            return (L.dil_ntt_dev(a_, NP, stream) | L.dil_ntt_dev(b_, NP, stream) | L.dil_pointwise_dev(a_, a_, b_, NP, stream)
                    | L.dil_invntt_dev(a_, NP, stream))
        c_ms, _ = timed(chain)
        sec.setdefault("other_configs", {})["polymul (ntt, ntt, pointwise, invntt) batch=32768"] = {
            "fused_products_per_s": NP / (f_ms * 1e-3), "fused_ms": f_ms, "bytes_per_product_fused": 3072,
            "fused_GBps": 3072 * NP / (f_ms * 1e-3) / 1e9, "fused_frac_of_hbm_peak": 3072 * NP / (f_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "four_launches_products_per_s": NP / (c_ms * 1e-3), "four_launches_ms": c_ms, "bytes_per_product_four_launches": 9216,
            "kernel": "polymul_kernel (both forward transforms, the product and the inverse transform of a pair in one wavefront's registers)"}
        del pa, pb, pc
    except Exception as e:  # noqa: BLE001
        sec.setdefault("other_configs", {})["polymul_error"] = repr(e)


def leg_scheme(cx, sec):
    """pk / sk / signature bytes in HBM -> bytes in HBM (level 3), rates at 8192 and 65536, and the
    whole-call latencies at batch 1 / 64 / 1024"""
    args, api, sharding, dlib, C, L, P, stream = cx.args, cx.api, cx.sharding, cx.dlib, cx.C, cx.L, cx.P, cx.stream
    rank, world, NS, hstreams, timed, with_clock, probe_effect = cx.rank, cx.world, cx.NS, cx.hstreams, cx.timed, cx.with_clock, cx.probe_effect
    # the whole scheme from wire bytes on the device (level 3, batch 8192 per GPU)
    try:
        g3 = torch.Generator(device="cuda").manual_seed(9 + rank)
        u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g3)  # noqa: E731
        seed, mu = u8(VBATCH, 32), u8(VBATCH, 64)
        pkb, skb, sgb = api.pk_bytes(3), api.sk_bytes(3), api.sig_bytes(3)
        pk = torch.empty((VBATCH, pkb), dtype=torch.uint8, device="cuda")
        sk = torch.empty((VBATCH, skb), dtype=torch.uint8, device="cuda")
        sig = torch.empty((VBATCH, sgb), dtype=torch.uint8, device="cuda")
        sigd = torch.empty((VBATCH, sgb), dtype=torch.uint8, device="cuda")
        att = torch.empty((VBATCH,), dtype=torch.int32, device="cuda")
        vd = torch.empty((VBATCH,), dtype=torch.int32, device="cuda")
        kg_ms, _ = timed(lambda i: L.dil_keygen_dev(P(pk), P(sk), P(seed), 3, VBATCH, stream))
        sg_ms, _ = timed(lambda i: L.dil_sign_dev(P(sig), P(att), P(sk), P(mu), 3, VBATCH, 1, 512, stream))
        # the same with the round's count fetched by copy + event (rounds 2-5's form, option sign_wake = 0)
        cx.api.set_option("sign_wake", 0)
        try:
            sg0_ms, _ = timed(lambda i: L.dil_sign_dev(P(sig), P(att), P(sk), P(mu), 3, VBATCH, 1, 512, stream))
        finally:
            cx.api.set_option("sign_wake", 1)
        mean_att = float(att.float().mean())
        sgd_ms, _ = timed(lambda i: L.dil_sign_dev(P(sigd), P(att), P(sk), P(mu), 3, VBATCH, 0, 512, stream))
        vf_ms, _ = timed(lambda i: L.dil_verify_sig_dev(P(vd), P(pk), P(sig), P(mu), 3, VBATCH, 1, stream))
        ok = int(vd.abs().sum()) == 0
        vfd_ms, _ = timed(lambda i: L.dil_verify_sig_dev(P(vd), P(pk), P(sigd), P(mu), 3, VBATCH, 0, stream))
        ok = ok and int(vd.abs().sum()) == 0
        # the fused wire-format kernel alone (A expanded beforehand): bytes = 30 KiB A + packed z, t1, hints, c, w1
        A3 = api.expand_a(pk[:, :32].contiguous(), 3)
        w1p = torch.empty((VBATCH, 6 * 128), dtype=torch.uint8, device="cuda")
        wk_ms, _ = timed(lambda i: L.dil_verify_wire_core_dev(P(w1p), P(vd), P(A3), P(pk), P(sigd), 3, VBATCH, 0, stream))
        # verification against keys whose matrix was expanded once and is kept across calls (dil_verify_sig_expanded_dev)
        vxd_ms, _ = timed(lambda i: L.dil_verify_sig_expanded_dev(P(vd), P(A3), P(pk), P(sigd), P(mu), 3, VBATCH, 0, stream))
        ok = ok and int(vd.abs().sum()) == 0
        vxs_ms, _ = timed(lambda i: L.dil_verify_sig_expanded_dev(P(vd), P(A3), P(pk), P(sig), P(mu), 3, VBATCH, 1, stream))
        ok = ok and int(vd.abs().sum()) == 0
        # ... and whose t1^ = NTT(t1 2^13) is kept too (dil_expand_t1_dev + dil_verify_sig_expanded2_dev): K forward transforms fewer per verification
        T3 = api.expand_t1(pk, 3)
        vx2_ms, _ = timed(lambda i: L.dil_verify_sig_expanded2_dev(P(vd), P(A3), P(T3), P(pk), P(sigd), P(mu), 3, VBATCH, 0, stream))
        ok = ok and int(vd.abs().sum()) == 0
        # the same at 8 x the batch (65536 per GPU): the latency-bound hash kernels are amortised
        BIG = 8 * VBATCH
        mu_b = u8(BIG, 64)
        sig_b = torch.empty((BIG, sgb), dtype=torch.uint8, device="cuda")
        att_b = torch.empty((BIG,), dtype=torch.int32, device="cuda")
        vd_b = torch.empty((BIG,), dtype=torch.int32, device="cuda")
        sgb_ms, _ = timed(lambda i: L.dil_sign_dev(P(sig_b), P(att_b), P(sk), P(mu_b), 3, BIG, 1, 512, stream))
        vfb_ms, _ = timed(lambda i: L.dil_verify_sig_dev(P(vd_b), P(pk), P(sig_b), P(mu_b), 3, BIG, 1, stream))
        ok = ok and int(vd_b.abs().sum()) == 0
        seed_b = u8(BIG, 32)
        pk_b = torch.empty((BIG, pkb), dtype=torch.uint8, device="cuda")
        sk_b = torch.empty((BIG, skb), dtype=torch.uint8, device="cuda")
        kgb_ms, _ = timed(lambda i: L.dil_keygen_dev(P(pk_b), P(sk_b), P(seed_b), 3, BIG, stream))
        dlib.check(L.dil_sign_dev(P(sig_b), P(att_b), P(sk_b), P(mu_b), 3, BIG, 0, 512, stream))
        vdb_ms, _ = timed(lambda i: L.dil_verify_sig_dev(P(vd_b), P(pk_b), P(sig_b), P(mu_b), 3, BIG, 0, stream))
        ok = ok and int(vd_b.abs().sum()) == 0
        # the operations on (key, message): mu = SHAKE256(tr || M) on the device too (64-byte messages, one key for the batch)
        blob = u8(VBATCH * 64)
        offs = (torch.arange(VBATCH, device="cuda", dtype=torch.int64) * 64).contiguous()
        lens = torch.full((VBATCH,), 64, dtype=torch.int32, device="cuda")
        sig_m = torch.empty((VBATCH, sgb), dtype=torch.uint8, device="cuda")
        sm_ms, _ = timed(lambda i: L.dil_sign_msg_dev(P(sig_m), P(att), P(sk), P(blob), blob.numel(), P(offs), P(lens), 3, VBATCH, 1, 512, stream))
        vm_ms, _ = timed(lambda i: L.dil_verify_msg_dev(P(vd), P(pk), P(sig_m), P(blob), blob.numel(), P(offs), P(lens), 3, VBATCH, 1, stream))
        ok = ok and int(vd.abs().sum()) == 0
        per_s = lambda ms: VBATCH / (ms * 1e-3)  # noqa: E731
        sec["scheme_level3_wire_format"] = {
            "note": "pk/sk/sig bytes in HBM -> bytes in HBM; SHAKE, samplers, codecs, rejection loop all on the device; "
                    "verification reads the packed fields inside the fused kernel (no int32 temporaries).  Rates are whole calls "
                    "timed with HIP events around back-to-back calls on one stream (the median of three regions of >= 25 ms); the sign rates are therefore HOST-INCLUSIVE: "
                    "dil_sign_dev sizes every rejection round on the host from the count the round's last kernel posts into mapped host words (option sign_wake)",
            "keygen_per_s": per_s(kg_ms), "sign_shared_key_per_s": per_s(sg_ms), "sign_shared_key_copy_event_per_s": per_s(sg0_ms),
            "sign_distinct_keys_per_s": per_s(sgd_ms),
            "verify_shared_pk_per_s": per_s(vf_ms), "verify_distinct_pk_per_s": per_s(vfd_ms),
            "verify_wire_core_distinct_pk": {"per_s": per_s(wk_ms), "ms": wk_ms, "kernel": "verify_wire_wpi_kernel<3> with SampleInBall "
                                             "inside (option fuse_sib; rounds 1-5: sample_in_ball_bits_kernel in front)", "wire_bytes_per_verify": 30 * 1024 + 3200 + 61 + 1920 + 256 + 768},
            "verify_expanded_keys": {"distinct_pk_per_s": per_s(vxd_ms), "shared_pk_per_s": per_s(vxs_ms),
                                     "distinct_pk_with_t1hat_per_s": per_s(vx2_ms),
                                     "note": "A = ExpandA(rho) expanded once by the caller and kept across calls"},
            "messages_64B_one_key": {"sign_msg_per_s": per_s(sm_ms), "verify_msg_per_s": per_s(vm_ms)},
            "mean_sign_attempts": mean_att, "all_signatures_verify": ok, "batch": VBATCH,
            "batch_65536": {"keygen_per_s": BIG / (kgb_ms * 1e-3), "sign_shared_key_per_s": BIG / (sgb_ms * 1e-3),
                            "verify_shared_pk_per_s": BIG / (vfb_ms * 1e-3), "verify_distinct_pk_per_s": BIG / (vdb_ms * 1e-3)}}
        # small-batch latency of one whole call (launch-bound): wall time per call incl. the host side
        lat = {}
        for nb in (1, 64, 1024):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                L.dil_verify_sig_dev(P(vd), P(pk), P(sigd), P(mu), 3, nb, 0, stream)
            torch.cuda.synchronize()
            lat[f"verify_sig_batch_{nb}_ms"] = (time.perf_counter() - t0) / reps * 1e3
        for name, call in (("keygen_batch_1_ms", lambda: L.dil_keygen_dev(P(pk), P(sk), P(seed), 3, 1, stream)),
                           ("sign_batch_1_ms", lambda: L.dil_sign_dev(P(sig), P(att), P(sk), P(mu), 3, 1, 1, 512, stream))):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                call()
            torch.cuda.synchronize()
            lat[name] = (time.perf_counter() - t0) / 20 * 1e3
        # the drop-in surface's batch-of-one call (libdil256_ref.so ntt() = dil_ntt_host(a, 1)): resident mailbox wave vs a launch
        # per call; host wall time per call, ctypes' ~1 us included in both
        import ctypes as _C
        one = np.arange(256, dtype=np.int32)
        onep = one.ctypes.data_as(_C.POINTER(_C.c_int32))
        for mode, key, reps in ((1, "ntt_host_batch_1_us", 5000), (0, "ntt_host_batch_1_launch_path_us", 300)):
            L.dil_set_option(b"host_mailbox", mode)
            for _ in range(20):
                L.dil_ntt_host(onep, 1)
            t0 = time.perf_counter()
            for _ in range(reps):
                L.dil_ntt_host(onep, 1)
            lat[key] = (time.perf_counter() - t0) / reps * 1e6
        L.dil_set_option(b"host_mailbox", 0)
        torch.cuda.synchronize()
        sec["scheme_level3_wire_format"]["latency"] = lat
    except Exception as e:  # noqa: BLE001
        sec["scheme_level3_wire_format"] = {"error": repr(e)}


def leg_end_to_end(cx):
    """"exclude H2D/D2H from kernel figures but report end-to-end separately".  The reference's calling convention for the
    path is caller-owned HOST arrays (reference_code/ref_ntt.h:30-36, hardware_code/ntt2x2.h:30-34); these are the same two workloads
    through the host-pointer entry points (csrc/capi.hip: chunks of H2D -> kernel -> D2H; a pageable buffer goes through the library's ring of
    page-locked slots by memcpy, a page-locked one is DMA'd in place with one stream per direction), with the caller's buffers pageable and page-locked, against the link's own copy rate
    measured here -- one direction at a time, and both at once in the best pattern found (scripts/bench_pcie_duplex.py).  PCIe-bound by two orders of magnitude: never `value`."""
    api = cx.api

    def med(f, reps=5):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    out = {}
    nbytes = 256 << 20
    hbuf = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dbuf = torch.empty(nbytes, dtype=torch.uint8, device="cuda")

    def h2d():
        dbuf.copy_(hbuf, non_blocking=True)
        torch.cuda.synchronize()

    def d2h():
        hbuf.copy_(dbuf, non_blocking=True)
        torch.cuda.synchronize()
    h2d(), d2h()
    hbuf2 = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dbuf2 = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()

    def both(chunk=8 << 20):
        for off in range(0, nbytes, chunk):
            with torch.cuda.stream(s_up):
                dbuf[off:off + chunk].copy_(hbuf[off:off + chunk], non_blocking=True)
            with torch.cuda.stream(s_dn):
                hbuf2[off:off + chunk].copy_(dbuf2[off:off + chunk], non_blocking=True)
        torch.cuda.synchronize()
    both()
    link = {"h2d_GBps": nbytes / med(h2d) / 1e9, "d2h_GBps": nbytes / med(d2h) / 1e9, "duplex_GBps_each_way": nbytes / med(both) / 1e9,
            "how": "256 MiB page-locked <-> device copies (torch), median of 5, one direction at a time; duplex = both directions at once as "
                   "8-MiB copies on one stream per direction, the best pattern found (two whole-buffer copies side by side share the "
                   "one-way rate: profiles/r05t_pcie_duplex.txt)"}
    del hbuf, dbuf, hbuf2, dbuf2
    out["pcie"] = link
    one_way = min(link["h2d_GBps"], link["d2h_GBps"])
    rng = np.random.default_rng(3)
    a = rng.integers(0, 8380417, (BATCH, 256), dtype=np.int32)
    ntt = {"workload": "BASELINE configs[1] through dil_ntt_host + dil_invntt_host: 65536 polynomials, 64 MiB up and 64 MiB down per call",
           "options": {k: api.get_option(k) for k in ("host_chunk", "host_streams", "host_copy_threads", "host_duplex")}}
    for kind in ("pageable", "page_locked"):
        keep = torch.from_numpy(a.copy()).pin_memory() if kind == "page_locked" else None
        x = keep.numpy() if keep is not None else a.copy()
        api.ntt(x), api.invntt(x)
        assert (x == a).all(), "host round trip is not the identity"
        tf, ti = med(lambda: api.ntt(x)), med(lambda: api.invntt(x))
        gb = BATCH * 1024 / ((tf + ti) / 2) / 1e9
        ntt[kind] = {"value": 2 * BATCH / (tf + ti), "unit": "NTT/s", "fwd_ms": tf * 1e3, "inv_ms": ti * 1e3,
                     "GBps_each_way": gb, "frac_of_pcie": gb / one_way, "frac_of_duplex_link": gb / link["duplex_GBps_each_way"],
                     "note": "frac_of_pcie = bytes one way / call time, over the slower direction's copy rate alone: 1.0 = both directions "
                             "fully overlapped at the one-way rate; frac_of_duplex_link = over what the link gave each way with both busy"}
    # the batch at which a host caller is better off here than on the CPU: per-call wall time of dil_ntt_host, pageable buffer
    sweep = {}
    for b in (1, 4, 16, 64, 256, 1024, 4096, 16384, 65536):
        xb = a[:b].copy()
        api.ntt(xb)
        reps = 200 if b <= 1024 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            api.ntt(xb)
        sweep[str(b)] = b / ((time.perf_counter() - t0) / reps)
    ntt["batch_sweep_NTT_per_s"] = sweep
    out["ntt"] = ntt
    # the one call that is worth a PCIe round trip for a source-compatible caller: the whole polymul chain in one upload + one download
    try:
        NPH = 32768
        ha, hb = a[:NPH].copy(), rng.integers(0, 8380417, (NPH, 256), dtype=np.int32)
        hc = np.empty_like(ha)
        api.polymul(hc, ha, hb)
        t_f = med(lambda: api.polymul(hc, ha, hb))

        def four_calls():
            x, y = ha.copy(), hb.copy()
            t0 = time.perf_counter()
            api.ntt(x), api.ntt(y)
            api.pointwise_barrett(x, x, y)
            api.invntt(x)
            return time.perf_counter() - t0
        four_calls()
        t_c = float(np.median([four_calls() for _ in range(3)]))
        out["polymul"] = {"workload": "32768 products from pageable host arrays: dil_polymul_host (a, b up once, c down) against the chain as four *_host calls",
                          "one_call": {"value": NPH / t_f, "unit": "products/s", "ms": t_f * 1e3, "bytes_up": 2048 * NPH, "bytes_down": 1024 * NPH,
                                       "GBps_up": 2048 * NPH / t_f / 1e9, "frac_of_pcie": 2048 * NPH / t_f / 1e9 / one_way},
                          "four_calls": {"value": NPH / t_c, "unit": "products/s", "ms": t_c * 1e3}}
    except Exception as e:  # noqa: BLE001
        out["polymul"] = {"error": repr(e)}
    # configs[3]: host A / z / c / t1 / h -> w1
    A, z, c, t1_, h = synth_verify(VBATCH, 901)
    h2 = np.ascontiguousarray(h.reshape(VBATCH, -1))
    up = A.nbytes + z.nbytes + c.nbytes + t1_.nbytes + h2.nbytes
    ver = {"workload": "BASELINE configs[3] through dil_verify_core_host: 8192 level-3 items, a key per item", "bytes_up": int(up),
           "bytes_down": int(VBATCH * 6 * 256)}
    for kind in ("pageable", "page_locked"):
        arrs = [A, z, c, t1_, h2]
        keep = None
        if kind == "page_locked":
            keep = [torch.from_numpy(x).pin_memory() for x in arrs]
            arrs = [k.numpy() for k in keep]
        w1 = api.verify_core(arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], 3)
        t = med(lambda: api.verify_core(arrs[0], arrs[1], arrs[2], arrs[3], arrs[4], 3, out=w1.reshape(VBATCH, -1)), 3)
        ver[kind] = {"value": VBATCH / t, "unit": "verify/s", "ms": t * 1e3, "GBps_up": up / t / 1e9, "frac_of_pcie": up / t / 1e9 / link["h2d_GBps"]}
    out["verify"] = ver
    return out


def bench_configs4_sharded(cx):
    """level-5 sign inner loop (phase 1 + phase 2, one signing key) and the whole signing loop on one batch of
    8192 x world items: every rank builds the SAME batch (same seed), run_sharded hands it its contiguous slice, the
    HIP kernels run on the slice, gather_slabs all-gathers the result slabs"""
    L, P, stream, rank, world, timed, sharding, api = cx.L, cx.P, cx.stream, cx.rank, cx.world, cx.timed, cx.sharding, cx.api
    per, K5, L5 = 8192, 8, 7
    total = per * world
    gq = torch.Generator(device="cuda").manual_seed(4242)          # same on every rank: one logical batch
    rnd = lambda *sh: torch.randint(0, 8380417, sh, dtype=torch.int32, device="cuda", generator=gq)  # noqa: E731
    small = lambda lo, hi, *sh: (torch.randint(lo, hi + 1, sh, dtype=torch.int64, device="cuda", generator=gq) % 8380417).to(torch.int32)  # noqa: E731
    A5 = rnd(1, K5, L5, 256)
    s1h, s2h, t0h = small(-2, 2, 1, L5, 256), small(-2, 2, 1, K5, 256), small(-4095, 4096, 1, K5, 256)   # eta = 2, d = 13
    for t in (s1h, s2h, t0h):
        api.ntt(t)
    g1 = 1 << 19
    y = small(-(g1 - 1), g1, total, L5, 256)
    cc = api.sample_in_ball(torch.randint(0, 256, (total, 32), dtype=torch.uint8, device="cuda", generator=gq), 5)
    out = {}

    def attempt(ys, cs):
        n = ys.shape[0]
        w1 = torch.empty((n, K5, 256), dtype=torch.uint8, device="cuda")
        w0 = torch.empty((n, K5, 256), dtype=torch.int32, device="cuda")
        z = torch.empty((n, L5, 256), dtype=torch.int32, device="cuda")
        h = torch.empty((n, K5, 256), dtype=torch.uint8, device="cuda")
        f = torch.empty((n,), dtype=torch.int32, device="cuda")

        def one(i):
            return L.dil_sign_phase1_dev(P(w1), P(w0), P(A5), P(ys), 5, n, 1, stream) | \
                L.dil_sign_phase2_skey_dev(P(z), P(h), P(f), P(cs), P(ys), P(w0), P(w1), P(s1h), P(s2h), P(t0h), 5, n, 1, 0, stream)
        ms, _ = timed(one)
        out["attempt_ms_per_rank_slice"] = sharding.max_over_ranks(ms)
        return z, h, f

    z, h, f = sharding.run_sharded(attempt, total, y, cc, gather=False)
    torch.cuda.synchronize()
    sharding.barrier()
    t0 = time.perf_counter()
    gz, gh, gf = (sharding.gather_slabs(t, total) for t in (z, h, f))
    torch.cuda.synchronize()
    gather_ms = (time.perf_counter() - t0) * 1e3
    assert gz.shape[0] == total and gf.shape[0] == total
    out.update({"workload": "BASELINE configs[4]: level 5 (K=8, L=7) sign inner loop, one key, batch = 8192 per GPU x "
                            f"{world} GPU(s) = {total}, contiguous slices (run_sharded), gather of z + h + flag slabs",
                "attempts_per_s": total / (out["attempt_ms_per_rank_slice"] * 1e-3),
                "final_gather_ms": sharding.max_over_ranks(gather_ms) if world > 1 else None,      # at N = 1 there is nothing to gather
                "final_gather_bytes": int(gz.numel() * 4 + gh.numel() + gf.numel() * 4),
                "final_gather_GBps_per_gpu_received": (None if world == 1 else
                                                       int(gz.numel() * 4 + gh.numel() + gf.numel() * 4) * (world - 1) / world
                                                       / (sharding.max_over_ranks(gather_ms) * 1e-3) / 1e9),
                "xgmi_bound_GBps_per_gpu": None if world == 1 else 153.0 * min(world - 1, 7),
                "xgmi_bound_source": "SURVEY.md 5: 7 xGMI links x ~153 GB/s per GPU, point to point",
                "accept_rate": float((gf == 0).float().mean())})
    # the whole signing loop (rejection sampling to completion) on the same sharding: KAT-style deterministic signatures
    gm = torch.Generator(device="cuda").manual_seed(777)
    seed = torch.randint(0, 256, (1, 32), dtype=torch.uint8, device="cuda", generator=gm)
    mu = torch.randint(0, 256, (total, 64), dtype=torch.uint8, device="cuda", generator=gm)
    pk, sk = api.keygen(seed, 5)

    def sign(mus):
        n = mus.shape[0]
        sig = torch.empty((n, api.sig_bytes(5)), dtype=torch.uint8, device="cuda")
        att = torch.empty((n,), dtype=torch.int32, device="cuda")
        ms, _ = timed(lambda i: L.dil_sign_dev(P(sig), P(att), P(sk), P(mus), 5, n, 1, 512, stream))
        out["sign_ms_per_rank_slice"] = sharding.max_over_ranks(ms)
        return sig
    sig = sharding.run_sharded(sign, total, mu, gather=False)
    torch.cuda.synchronize()
    sharding.barrier()
    t0 = time.perf_counter()
    gsig = sharding.gather_slabs(sig, total)
    torch.cuda.synchronize()
    out["signatures_per_s"] = total / (out["sign_ms_per_rank_slice"] * 1e-3)
    sg_ms_ = sharding.max_over_ranks((time.perf_counter() - t0) * 1e3)
    out["signature_gather_ms"] = sg_ms_ if world > 1 else None
    out["signature_gather_bytes"] = int(gsig.numel())
    lo = 0 if world == 1 else (total // 2)
    vd = api.verify_sig(pk, gsig[lo:lo + 2048].contiguous(), mu[lo:lo + 2048].contiguous(), 5, shared_pk=True)
    out["gathered_signatures_verify"] = int(vd.abs().sum()) == 0
    return out



def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--prewarm-ms", type=float, default=200.0,
                    help="untimed: keep launching steps for this long before the W warm-up steps, so that the GPU has "
                         "left its idle power state (short runs measured 10 %% low without it)")
    ap.add_argument("--rotate", type=int, default=8, help="distinct resident batches the steps rotate over")
    ap.add_argument("--streams", type=int, default=2,
                    help="HIP streams the steps alternate over (step i runs on stream i %% S; a batch always stays on one "
                         "stream).  2 keeps a second launch in flight, which fills the dispatch gap and the ramp/tail of "
                         "every kernel: +15 %% over one stream")
    ap.add_argument("--region-ms", type=float, default=50.0,
                    help="a timed region of the headline is a whole number of K-step groups at least this long")
    ap.add_argument("--regions", type=int, default=7, help="timed regions of the headline (at least)")
    ap.add_argument("--total-ms", type=float, default=400.0, help="the headline's regions add up to at least this much")
    ap.add_argument("--min-ms", type=float, default=25.0,
                    help="every secondary leg is timed for at least this long, whatever --steps says")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-verify-overlap", action="store_true",
                    help="skip the two-stream leg of the secondary verify core (profiling runs: keeps rocprofv3's average "
                         "duration of verify_wpi_kernel a single-kernel figure)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-pointer (PCIe-inclusive) legs")
    return ap.parse_args()


def main():
    args = parse_args()
    cx = Bench(args)
    rank, world = cx.rank, cx.world
    out = leg_headline(cx)
    if world > 1:
        leg_distributed(cx, out)
    if not args.no_secondary:
        # (a secondary leg that fails leaves its error in the line: the headline above is printed whatever happens below)
        try:
            sec = leg_verify_core(cx)
        except Exception as e:  # noqa: BLE001
            sec = {"error": repr(e)}
        for name, leg in (("other_configs", leg_other_configs), ("scheme_level3_wire_format", leg_scheme)):
            try:
                leg(cx, sec)
            except Exception as e:  # noqa: BLE001
                sec[name + "_error"] = repr(e)
        # BASELINE configs[4] as north_star describes it: ONE batch of level-5 signing work, 8192 items per GPU, sharded by contiguous
        # slices over the ranks (sharding.run_sharded), HIP compute on every rank, then the one collective of the design: the gather of
        # the (z, h, flag) result slabs over RCCL/xGMI -- timed separately.
        try:
            sec["configs4_sharded"] = bench_configs4_sharded(cx)
        except Exception as e:  # noqa: BLE001
            sec["configs4_sharded"] = {"error": repr(e)}
        out["secondary"] = sec
    if rank == 0 and world == 1 and not args.no_end_to_end:
        try:
            out["end_to_end"] = leg_end_to_end(cx)
        except Exception as e:  # noqa: BLE001
            out["end_to_end"] = {"error": repr(e)}
    if not args.no_secondary and "roofline" in out.get("secondary", {}):
        # the second half of BASELINE.json's metric (Dilithium-3 verifies/s) inside the blocks the driver's record keeps: `roofline.verify_core`
        # (+ the same scalars flat, `verify_core_*`, should a reader keep only scalars); op-graph: rtl_src/combined_top.v:1207-1469
        sec = out["secondary"]
        vr = sec["roofline"]
        vc = {"metric": sec["metric"], "value": sec["value"], "unit": sec["unit"], "kernel": vr["kernel"], "bound": vr["bound"],
              "achieved": vr["achieved"], "peak": vr["peak"], "frac": vr["frac"], "avg_launch_ms": vr["avg_launch_ms"],
              "algorithmic_bytes_per_launch": VERIFY3_BYTES * VBATCH, "traffic": vr.get("traffic"),
              "rotating_input_sets": sec["config"]["rotating_input_sets"], "batch": VBATCH,
              "workload": "BASELINE configs[3]: level-3 verify core, batch 8192, a key per item"}
        out["roofline"]["verify_core"] = vc
        for k in ("value", "frac", "achieved", "avg_launch_ms", "traffic", "rotating_input_sets"):
            out["roofline"]["verify_core_" + k] = vc[k]
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            cb = out["cpu_baseline"]
            try:
                cb["all_threads"] = cpu_baseline_all_threads(1.0 / cb["value"])
                cb["all_threads_value"], cb["all_threads_cores"] = cb["all_threads"]["value"], cb["all_threads"]["cores"]
            except Exception as e:  # noqa: BLE001
                cb["all_threads"] = {"error": repr(e)}
            if not args.no_secondary:
                legs = (("verify_core", lambda: dict(cpu_baseline_verify(), all_threads=cpu_baseline_verify_all_threads())),
                        ("matvec", cpu_baseline_matvec), ("sign_attempt", cpu_baseline_sign_attempt))
                for name, leg in legs:       # CPU figures of the other BASELINE configs (every GPU rate has its host rate beside it)
                    try:
                        cb[name] = leg()
                        cb[name + "_value"] = cb[name]["value"]
                        cb[name + "_all_threads_value"] = cb[name]["all_threads"]["value"]
                    except Exception as e:  # noqa: BLE001
                        cb[name] = {"error": repr(e)}
                out["secondary"]["cpu_baseline"] = cb.get("verify_core")
            sweep = out.get("end_to_end", {}).get("ntt", {}).get("batch_sweep_NTT_per_s")
            if sweep:       # the batch from which one dil_ntt_host call beats the CPU reference on this box
                first = lambda rate: next((int(b) for b, v in sweep.items() if v > rate), None)  # noqa: E731
                allt = out["cpu_baseline"].get("all_threads", {}).get("value")
                out["end_to_end"]["ntt"]["crossover_batch"] = {"vs_one_cpu_thread": first(out["cpu_baseline"]["value"]),
                                                               "vs_all_cpu_threads": first(allt) if allt else None}
        print(json.dumps(out))
    cx.sharding.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
