"""The CPU legs of bench.py (the `cpu_baseline` object) run without a GPU and stay bounded."""
import time

import bench


def test_cpu_baseline_single_thread_is_bounded():
    t0 = time.perf_counter()
    b = bench.cpu_baseline(sample_polys=256, target_s=0.5)
    assert time.perf_counter() - t0 < 20
    assert b["unit"] == "NTT/s" and b["cores"] == 1 and b["kind"] in ("reference", "port")
    assert 1e4 < b["value"] < 1e8


def test_cpu_baseline_all_threads_is_time_bounded():
    t0 = time.perf_counter()
    b = bench.cpu_baseline_all_threads(2e-6, target_s=1.0, sample_polys=256)
    assert time.perf_counter() - t0 < 20          # bounded by the deadline, whatever CPU quota the box really has
    assert b["cores"] >= 1 and b["value"] > 1e4


def test_cpu_baseline_verify():
    b = bench.cpu_baseline_verify(target_s=0.5)
    assert b["unit"] == "verify/s" and b["value"] > 100
