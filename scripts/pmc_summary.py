#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*) into profiles/pmc_summary.json.

usage: pmc_summary.py <out.json> <results.db> [<results.db> ...]

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read, so
hbm_read_bytes = 2 * FETCH_SIZE * 1024.  WRITE_SIZE is taken as is (calibrated here on the
in-place NTT, which writes exactly 64 MiB per launch, and on verify's 12 MiB of w1)."""
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the sources a kernel family is compiled from: bench.py refuses a committed summary whose stamps no longer match the tree
SOURCES = {
    "ntt": ["kernels.hip", "ntt_core.hpp", "modarith.hpp", "device_common.hpp"],
    "verify": ["pipelines.hip", "pipeline_common.hpp", "ntt_core.hpp", "modarith.hpp", "device_common.hpp"],
    "sign": ["pipelines.hip", "pipeline_common.hpp", "ntt_core.hpp", "modarith.hpp", "device_common.hpp"],
}


def git_blob_id(path):
    """what `git hash-object` prints for the file"""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_stamps():
    csrc = os.path.join(ROOT, "dilithium_amd", "csrc")
    return {fam: {f: git_blob_id(os.path.join(csrc, f)) for f in files} for fam, files in SOURCES.items()}


def main():
    out = {}
    for path in sys.argv[2:]:
        cur = sqlite3.connect(path).cursor()
        rows = cur.execute(
            "select name, counter_name, avg(v), count(*) from (select name, counter_name, dispatch_id, "
            "sum(counter_value) as v from pmc_events group by dispatch_id, counter_name) group by name, counter_name").fetchall()
        for name, cn, v, n in rows:
            if "dil::" not in name:
                continue
            key = name.split("dil::")[1].split("(")[0]           # e.g. ntt_fwd_kernel<0>
            out.setdefault(key, {})[cn] = v
            out[key]["dispatches"] = n
        for name, d in cur.execute("select name, avg(end-start) from kernels group by name").fetchall():
            if "dil::" in name:
                key = name.split("dil::")[1].split("(")[0]
                out.setdefault(key, {}).setdefault("avg_ns_under_pmc", d)
    for key, d in out.items():
        if "FETCH_SIZE" in d or "WRITE_SIZE" in d:
            rd = 2.0 * d.get("FETCH_SIZE", 0.0) * 1024.0
            wr = d.get("WRITE_SIZE", 0.0) * 1024.0
            d["hbm_read_bytes_per_launch"] = rd
            d["hbm_write_bytes_per_launch"] = wr
            d["hbm_bytes_per_launch"] = rd + wr
    base = lambda k: k.split("<")[0]  # noqa: E731
    # convenience aliases used by bench.py
    for key in list(out):
        if key.startswith("ntt_fwd_kernel<0>"):
            out["ntt_fwd_kernel"] = out[key]
        if key.startswith("ntt_inv_kernel<0>"):
            out["ntt_inv_kernel"] = out[key]
        if base(key) in ("verify_wpi_kernel", "verify_kernel") and "<3>" in key:
            out["verify_kernel"] = out[key]
        if base(key) == "verify_wire_wpi_kernel" and "<3>" in key:
            out["verify_wire_kernel"] = out[key]
    if os.path.exists(sys.argv[1]):                 # keys other passes put there (sign_valu_insts_per_attempt) survive a refresh of these
        try:
            old = json.load(open(sys.argv[1]))
            for k in ("sign_valu_insts_per_attempt",):
                if k in old and k not in out:
                    out[k] = old[k]
        except Exception:
            pass
    out["_source_blobs"] = source_stamps()
    json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    for k, d in sorted(out.items()):
        if isinstance(d, dict) and "hbm_bytes_per_launch" in d:
            print(f"{k:28s} read {d['hbm_read_bytes_per_launch'] / 1e6:9.1f} MB  write {d['hbm_write_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
