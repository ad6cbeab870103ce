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
  roofline      dominant kernels (forward / inverse NTT, alternating, same bytes): algorithmic bytes
                (2048 B x 65536 per launch) / the average launch duration over the timed region, from
                two HIP events on the launch stream around the region (no events inside it: an event
                record between launches costs microseconds of GPU idle time); a separate untimed
                instrumented pass splits forward from inverse.  peak 8 TB/s (MI355X_MICROARCH.md);
                traffic from the committed rocprofv3 PMC pass (profiles/), or null.
  cpu_baseline  the reference's own ntt()+invntt() (oracle/_ref, kind "reference") -- or our C
                restatement (kind "port") -- on one host core, bounded sample.
  secondary     Dilithium-3 verify cores / s (configs[3], batch 8192, distinct pk) with its own
                roofline fraction; and the final RCCL gather time when N > 1.
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
    return {"value": n / dt, "unit": "NTT/s", "cores": nthreads,
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


def pmc_traffic(kernel_key):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/), if any"""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if not os.path.exists(p):
        return None
    try:
        return json.load(open(p)).get(kernel_key, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def main():
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
    ap.add_argument("--verify-streams", type=int, default=1, help="streams of the secondary (fused verify) metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    from dilithium_amd import api, sharding
    from dilithium_amd import lib as dlib
    import ctypes as C

    rank, world, local = sharding.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    api.init(local)
    L = dlib.load()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def ev():
        e = C.c_void_p()
        dlib.check(L.dil_event_create(C.byref(e)))
        return e

    def elapsed(a, b):
        ms = C.c_float()
        dlib.check(L.dil_event_elapsed_ms(C.byref(ms), a, b))
        return float(ms.value)

    # ---- inputs, resident in HBM ------------------------------------------------------------
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    NS = max(1, args.streams)
    R = max(1, args.rotate)
    R += (-R) % NS                                        # a batch must always meet the same stream
    bufs = [torch.randint(0, 8380417, (BATCH, 256), dtype=torch.int32, device="cuda", generator=g) for _ in range(R)]
    check = bufs[0][:64].clone()
    torch.cuda.synchronize()

    ptrs = [C.c_void_p(b.data_ptr()) for b in bufs]      # direct C-ABI calls: minimal host overhead
    tstreams = [torch.cuda.Stream() for _ in range(NS)] if NS > 1 else [torch.cuda.current_stream()]
    hstreams = [C.c_void_p(ts.cuda_stream) for ts in tstreams]

    def step(i, fixed=None):
        p = ptrs[(i % R) if fixed is None else (fixed + i % NS)]
        st = hstreams[i % NS]
        return L.dil_ntt_dev(p, BATCH, st) | L.dil_invntt_dev(p, BATCH, st)

    def region(k, fixed=None):
        """EXACTLY k steps, bracketed by one HIP event pair per stream; returns (wall seconds, per-stream event ms)"""
        e0 = [ev() for _ in range(NS)]
        e1 = [ev() for _ in range(NS)]
        sharding.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = 0
        for j in range(NS):
            rc |= L.dil_event_record(e0[j], hstreams[j])
        for i in range(k):
            rc |= step(i, fixed)
        for j in range(NS):
            rc |= L.dil_event_record(e1[j], hstreams[j])
        torch.cuda.synchronize()
        sharding.barrier()
        wall = time.perf_counter() - t0
        dlib.check(rc, "timed NTT launches")
        return sharding.max_over_ranks(wall), [elapsed(a, b) for a, b in zip(e0, e1)]

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:      # untimed clock/power warm-up
        for i in range(32):
            step(i)
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    K = args.steps
    # Timed region: no events inside (an event record between two launches costs several microseconds of GPU idle
    # time -- the marker packet drains the pipeline -- which at 26 us per kernel is a 10 % perturbation).
    dt, ev_ms = region(K)
    assert torch.equal(bufs[0][:64], check), "fwd+inv round trip is not the identity"
    value = world * K * 2 * BATCH / dt
    launches = [2 * len(range(j, K, NS)) for j in range(NS)]            # per stream
    # average duration of one launch as the stream (and rocprof) sees it: with NS streams NS launches overlap
    launch_ms = float(np.mean([m / n for m, n in zip(ev_ms, launches) if n]))
    region_ms = max(ev_ms)
    ntt_gbs = NTT_BYTES * BATCH * 2 * K / (region_ms * 1e-3) / 1e9      # aggregate over the timed region

    # instrumented pass (NOT timed for `value`): one stream, events around single launches split forward from inverse;
    # each figure includes the event/dispatch overhead the timed region does not pay
    s_one = hstreams[0]
    ni = max(8, min(40, K // 4))
    evs = [(ev(), ev(), ev()) for _ in range(ni)]
    for i, (e0, e1, e2) in enumerate(evs):
        p = ptrs[(i * NS) % R]
        L.dil_event_record(e0, s_one)
        L.dil_ntt_dev(p, BATCH, s_one)
        L.dil_event_record(e1, s_one)
        L.dil_invntt_dev(p, BATCH, s_one)
        L.dil_event_record(e2, s_one)
    torch.cuda.synchronize()
    fwd_ms = float(np.mean([elapsed(e0, e1) for e0, e1, _ in evs]))
    inv_ms = float(np.mean([elapsed(e1, e2) for _, e1, e2 in evs]))
    # the same K steps on ONE stream, for reference
    one_stream_value = None
    if NS > 1:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(K):
            p = ptrs[i % R]
            L.dil_ntt_dev(p, BATCH, s_one)
            L.dil_invntt_dev(p, BATCH, s_one)
        torch.cuda.synchronize()
        one_stream_value = world * K * 2 * BATCH / sharding.max_over_ranks(time.perf_counter() - t1)

    # LLC-resident variant (the same NS batches every step: 64 MiB each, inside the 256 MiB Infinity Cache) for context
    for i in range(8):
        step(i, fixed=0)
    torch.cuda.synchronize()
    llc_dt, _ = region(K, fixed=0)
    llc_value = world * K * 2 * BATCH / llc_dt

    out = {
        "metric": "ntt256_transforms_per_sec", "value": value, "unit": "NTT/s", "n_gpus": world, "steps": K,
        "warmup": args.warmup, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: batched forward+inverse NTT, n=256, q=8380417, "
                               "batch=65536 polynomials per GPU; step = 1 fwd launch + 1 inv launch",
                   "batch_per_gpu": BATCH, "rotating_resident_batches": R, "streams": NS,
                   "parallelism": f"shard x{world}", "bytes_per_transform": NTT_BYTES},
        "roofline": {"bound": "hbm",
                     "kernel": "ntt_fwd_kernel<LAYOUT_POLY> / ntt_inv_kernel<LAYOUT_POLY> (alternating launches of the "
                               "timed region, identical algorithmic bytes)",
                     "achieved": ntt_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ntt_gbs / HBM_PEAK_GBS,
                     "traffic": pmc_traffic("ntt_fwd_kernel"), "avg_launch_ms": launch_ms,
                     "concurrent_launches": NS,
                     "algorithmic_bytes_per_launch": NTT_BYTES * BATCH,
                     "timing": "one HIP event pair per launch stream around the whole timed region; avg_launch_ms = that "
                               "stream's elapsed time / its launches (the per-kernel duration rocprofv3 reports); "
                               "achieved = all launches' algorithmic bytes / the region's elapsed time "
                               "(= concurrent_launches x bytes per launch / avg_launch_ms)",
                     "per_kernel_instrumented": {
                         "note": "separate untimed pass, events around single launches (adds event/dispatch overhead)",
                         "ntt_fwd_kernel_ms": fwd_ms, "ntt_inv_kernel_ms": inv_ms,
                         "ntt_fwd_kernel_GBps": NTT_BYTES * BATCH / (fwd_ms * 1e-3) / 1e9,
                         "ntt_inv_kernel_GBps": NTT_BYTES * BATCH / (inv_ms * 1e-3) / 1e9}},
        "llc_resident_value": llc_value,
        "one_stream_value": one_stream_value,
        # context: the same kernel without its loads/stores, i.e. the integer-ALU ceiling of this arithmetic
        "valu_ceiling": {"value": 3.74e9, "unit": "NTT/s per GPU", "source": "profiles/r01_tune_ntt.txt, compute-only variant "
                                                                             "(not measured in this run)"},
    }

    # ---- secondary: Dilithium-3 verify core, configs[3] ----------------------------------------
    if not args.no_secondary:
        cu = lambda x: torch.from_numpy(x).cuda()  # noqa: E731
        # The fused kernel holds its items' vectors in LDS and sizes its persistent grid to fill every CU, so a second
        # launch in flight cannot become resident beside it: measured 108 M/s on two streams vs 119 M/s on one -> one stream.
        VNS = max(1, min(NS, args.verify_streams))
        VSETS = max(2, VNS)          # rotate over >= 2 input sets (2 x 360 MiB > the 256 MiB Infinity Cache): HBM-streaming
        vsets = []
        for j in range(VSETS):
            A, z, c, t1_, h = synth_verify(VBATCH, 77 + rank + 100 * j)
            vsets.append((cu(A), cu(z), cu(c), cu(t1_), cu(h), torch.empty((VBATCH, 6, 256), dtype=torch.uint8, device="cuda")))
        dA, dz, dc, dt1, dh, w1 = vsets[0]
        vptr = [[C.c_void_p(t.data_ptr()) for t in vs_] for vs_ in vsets]
        torch.cuda.synchronize()

        def vstep(i):
            pA, pz, pc, pt1, ph, pw1 = vptr[i % VSETS]
            return L.dil_verify_core_dev(pw1, pA, pz, pc, pt1, ph, 3, VBATCH, 0, hstreams[i % VNS])

        vs = max(10, K // 4)
        for i in range(2 * VNS + 2):
            vstep(i)
        torch.cuda.synchronize()
        sharding.barrier()
        ve0 = [ev() for _ in range(VNS)]
        ve1 = [ev() for _ in range(VNS)]
        tv = time.perf_counter()
        rcv = 0
        for j in range(VNS):
            rcv |= L.dil_event_record(ve0[j], hstreams[j])
        for i in range(vs):
            rcv |= vstep(i)
        for j in range(VNS):
            rcv |= L.dil_event_record(ve1[j], hstreams[j])
        torch.cuda.synchronize()
        sharding.barrier()
        tv = sharding.max_over_ranks(time.perf_counter() - tv)
        dlib.check(rcv, "timed verify launches")
        v_ev = [elapsed(a_, b_) for a_, b_ in zip(ve0, ve1)]
        v_launches = [len(range(j, vs, VNS)) for j in range(VNS)]
        v_ms = float(np.mean([m / n for m, n in zip(v_ev, v_launches) if n]))     # per-kernel duration (NS overlap)
        v_gbs = VERIFY3_BYTES * VBATCH * vs / (max(v_ev) * 1e-3) / 1e9            # aggregate over the region
        sec = {"metric": "dilithium3_verify_cores_per_sec", "value": world * vs * VBATCH / tv, "unit": "verify/s",
               "config": {"workload": "BASELINE configs[3]: level-3 verify core (NTT z, A.z - c.t1.2^d, INTT, "
                                      "UseHint -> w1), batch=8192 per GPU, distinct pk (A, t1 per item)",
                          "bytes_per_verify": VERIFY3_BYTES, "streams": VNS, "rotating_input_sets": VSETS},
               "roofline": {"bound": "hbm", "kernel": "verify_wpi_kernel<3>", "achieved": v_gbs, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": v_gbs / HBM_PEAK_GBS, "traffic": pmc_traffic("verify_kernel"),
                            "avg_launch_ms": v_ms, "concurrent_launches": VNS}}
        # the same launches over ONE input set (360 MiB, partly served by the 256 MiB Infinity Cache), for context
        torch.cuda.synchronize()
        ea, eb = ev(), ev()
        L.dil_event_record(ea, hstreams[0])
        for i in range(vs):
            pA, pz, pc, pt1, ph, pw1 = vptr[0]
            L.dil_verify_core_dev(pw1, pA, pz, pc, pt1, ph, 3, VBATCH, 0, hstreams[0])
        L.dil_event_record(eb, hstreams[0])
        torch.cuda.synchronize()
        sec["llc_assisted_value"] = world * VBATCH / (elapsed(ea, eb) / vs * 1e-3)
        # same pipeline with ONE public key for the whole batch (A, t1 staged in LDS): VALU-bound, reported beside it
        torch.cuda.synchronize()
        e2, e3 = ev(), ev()
        L.dil_event_record(e2, stream)
        for _ in range(vs):
            api.verify_core(dA[:1], dz, dc, dt1[:1], dh, 3, shared_pk=True, out=w1)
        L.dil_event_record(e3, stream)
        torch.cuda.synchronize()
        s_ms = elapsed(e2, e3) / vs
        sec["shared_pk"] = {"value": VBATCH / (s_ms * 1e-3), "unit": "verify/s per GPU", "avg_launch_ms": s_ms,
                            "bytes_per_verify": 15 * 1024, "kernel": "verify_shared_kernel<3,16>",
                            "bound": "valu (key material LDS-resident)"}
        # configs[2] and configs[4] of BASELINE.json (parity-test configs; timed here for the record only)
        try:
            g2 = torch.Generator(device="cuda").manual_seed(5)
            rnd = lambda *sh: torch.randint(0, 8380417, sh, dtype=torch.int32, device="cuda", generator=g2)  # noqa: E731
            A2, y2 = rnd(4096, 4, 4, 256), rnd(4096, 4, 256)
            wout = torch.empty((4096, 4, 256), dtype=torch.int32, device="cuda")
            for _ in range(3):
                api.matvec(A2, y2, 2, out=wout)
            e4, e5 = ev(), ev()
            L.dil_event_record(e4, stream)
            for _ in range(vs):
                api.matvec(A2, y2, 2, out=wout)
            L.dil_event_record(e5, stream)
            torch.cuda.synchronize()
            m_ms = elapsed(e4, e5) / vs
            A5, y5, c5 = rnd(1, 8, 7, 256), rnd(8192, 7, 256), rnd(8192, 256)
            s1h, s2h, t0h = rnd(1, 7, 256), rnd(1, 8, 256), rnd(1, 8, 256)
            w1s, w0s = api.sign_phase1(A5, y5, 5, shared_key=True)
            for _ in range(2):
                api.sign_phase1(A5, y5, 5, shared_key=True)
                api.sign_phase2(c5, y5, w0s, w1s, s1h, s2h, t0h, 5, shared_key=True)
            e6, e7 = ev(), ev()
            L.dil_event_record(e6, stream)
            for _ in range(vs):
                api.sign_phase1(A5, y5, 5, shared_key=True)
                api.sign_phase2(c5, y5, w0s, w1s, s1h, s2h, t0h, 5, shared_key=True)
            L.dil_event_record(e7, stream)
            torch.cuda.synchronize()
            a_ms = elapsed(e6, e7) / vs
            sec["other_configs"] = {
                "configs[2] level-2 A.y matvec batch=4096 distinct A": {"matvecs_per_s": 4096 / (m_ms * 1e-3), "ms": m_ms,
                                                                         "GBps": 24 * 1024 * 4096 / (m_ms * 1e-3) / 1e9},
                "configs[4] level-5 sign attempt (phase1+phase2) batch=8192 per GPU, shared key": {
                    "attempts_per_s": 8192 / (a_ms * 1e-3), "ms": a_ms}}
        except Exception as e:  # noqa: BLE001
            sec["other_configs"] = {"error": repr(e)}
        # the whole scheme from wire bytes on the device (level 3, batch 8192 per GPU)
        try:
            g3 = torch.Generator(device="cuda").manual_seed(9 + rank)
            u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g3)  # noqa: E731
            seed, mu = u8(VBATCH, 32), u8(VBATCH, 64)

            def ev_time(fn, reps):
                fn()
                ea, eb = ev(), ev()
                L.dil_event_record(ea, stream)
                for _ in range(reps):
                    fn()
                L.dil_event_record(eb, stream)
                torch.cuda.synchronize()
                return elapsed(ea, eb) / reps

            kg_ms = ev_time(lambda: api.keygen(seed, 3), 5)
            pk, sk = api.keygen(seed, 3)
            sg_ms = ev_time(lambda: api.sign(sk[:1], mu, 3, shared_sk=True), 3)
            sig, att = api.sign(sk[:1], mu, 3, shared_sk=True)
            sgd_ms = ev_time(lambda: api.sign(sk, mu, 3), 2)
            vf_ms = ev_time(lambda: api.verify_sig(pk[:1], sig, mu, 3, shared_pk=True), 5)
            sigd, _ = api.sign(sk, mu, 3)
            vfd_ms = ev_time(lambda: api.verify_sig(pk, sigd, mu, 3), 5)
            ok = int(api.verify_sig(pk, sigd, mu, 3).abs().sum()) == 0 and \
                int(api.verify_sig(pk[:1], sig, mu, 3, shared_pk=True).abs().sum()) == 0
            # the same at 8 x the batch (65536 per GPU): the latency-bound hash kernels are amortised
            BIG = 8 * VBATCH
            mu_b = u8(BIG, 64)
            sgb_ms = ev_time(lambda: api.sign(sk[:1], mu_b, 3, shared_sk=True), 2)
            sig_b, _ = api.sign(sk[:1], mu_b, 3, shared_sk=True)
            vfb_ms = ev_time(lambda: api.verify_sig(pk[:1], sig_b, mu_b, 3, shared_pk=True), 3)
            ok = ok and int(api.verify_sig(pk[:1], sig_b, mu_b, 3, shared_pk=True).abs().sum()) == 0
            per_s = lambda ms: VBATCH / (ms * 1e-3)  # noqa: E731
            sec["scheme_level3_wire_format"] = {
                "note": "pk/sk/sig bytes in HBM -> bytes in HBM; SHAKE, samplers, codecs, rejection loop all on the device",
                "keygen_per_s": per_s(kg_ms), "sign_shared_key_per_s": per_s(sg_ms), "sign_distinct_keys_per_s": per_s(sgd_ms),
                "verify_shared_pk_per_s": per_s(vf_ms), "verify_distinct_pk_per_s": per_s(vfd_ms),
                "mean_sign_attempts": float(att.float().mean()), "all_signatures_verify": ok, "batch": VBATCH,
                "batch_65536": {"sign_shared_key_per_s": BIG / (sgb_ms * 1e-3), "verify_shared_pk_per_s": BIG / (vfb_ms * 1e-3)}}
        except Exception as e:  # noqa: BLE001
            sec["scheme_level3_wire_format"] = {"error": repr(e)}
        # the one collective of the design: final gather of the result slabs over RCCL/xGMI
        if world > 1:
            torch.cuda.synchronize()
            tg = time.perf_counter()
            allw1 = sharding.gather_slabs(w1, world * VBATCH)
            torch.cuda.synchronize()
            sec["final_gather_ms"] = (time.perf_counter() - tg) * 1e3
            sec["final_gather_bytes"] = int(allw1.numel())
        out["secondary"] = sec

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            try:
                out["cpu_baseline"]["all_threads"] = cpu_baseline_all_threads(1.0 / out["cpu_baseline"]["value"])
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"]["all_threads"] = {"error": repr(e)}
            if not args.no_secondary:
                out["secondary"]["cpu_baseline"] = cpu_baseline_verify()
        print(json.dumps(out))
    sharding.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
