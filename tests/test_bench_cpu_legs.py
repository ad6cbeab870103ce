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


def test_cpu_baseline_matvec_and_sign_attempt_are_bounded():
    """the host figures of BASELINE configs[2] / configs[4] (every GPU rate has its host rate beside it)"""
    t0 = time.perf_counter()
    m = bench.cpu_baseline_matvec(target_s=0.3)
    s = bench.cpu_baseline_sign_attempt(target_s=0.3)
    assert time.perf_counter() - t0 < 40
    assert m["unit"] == "matvec/s" and m["cores"] == 1 and m["kind"] == "port" and 1e3 < m["value"] < 1e7
    assert s["unit"] == "attempt/s" and s["cores"] == 1 and 1e2 < s["value"] < 1e6
    for b in (m, s):
        assert b["all_threads"]["threads"] >= 1 and b["all_threads"]["value"] > 0


def test_committed_pmc_summary_is_refused_when_stale(tmp_path, monkeypatch):
    """roofline.traffic comes from a committed rocprofv3 PMC summary: it carries the git blob ids of the kernel sources it was measured on
    (scripts/pmc_summary.py) and bench.py drops it the moment one of them changes"""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(bench.ROOT, "scripts"))
    import pmc_summary
    prof = tmp_path / "profiles"
    prof.mkdir()
    stamps = pmc_summary.source_stamps()
    good = {"ntt_fwd_kernel": {"hbm_bytes_per_launch": 134e6}, "verify_kernel": {"hbm_bytes_per_launch": 377e6}, "_source_blobs": stamps}
    (prof / "pmc_summary.json").write_text(json.dumps(good))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(pmc_summary, "ROOT", bench.ROOT if False else pmc_summary.ROOT)      # the stamps are always those of the real tree
    assert bench.pmc_traffic("ntt_fwd_kernel") == 134e6 and bench.pmc_traffic("verify_kernel") == 377e6
    stale = json.loads(json.dumps(good))
    stale["_source_blobs"]["ntt"]["kernels.hip"] = "0" * 40
    (prof / "pmc_summary.json").write_text(json.dumps(stale))
    assert bench.pmc_traffic("ntt_fwd_kernel") is None and bench.pmc_traffic("verify_kernel") == 377e6
    del stale["_source_blobs"]
    (prof / "pmc_summary.json").write_text(json.dumps(stale))
    assert bench.pmc_traffic("verify_kernel") is None
