#!/usr/bin/env python3
"""Which __global__ functions of the library did a traced run reach?  Reads a rocprofv3 rocpd database
(kernel trace of `pytest -m gpu`) and the kernel names declared in dilithium_amd/csrc/*.hip, prints every kernel
template with the instantiations and launch sizes seen, and lists the ones never launched.
usage: kernel_coverage.py <results.db>"""
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    names = {}
    for f in sorted(os.listdir(os.path.join(ROOT, "dilithium_amd", "csrc"))):
        if not f.endswith(".hip"):
            continue
        src = open(os.path.join(ROOT, "dilithium_amd", "csrc", f)).read()
        for m in re.finditer(r"__global__.*?\bvoid\s+(\w+)\s*\(", src, re.S):
            names[m.group(1)] = f
    return names


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), min(grid_x / workgroup_x), max(grid_x / workgroup_x), max(workgroup_x), "
                      "avg(end - start) from kernels group by name").fetchall()
    decl = declared()
    seen = {}
    for name, calls, gmin, gmax, wg, avg in rows:
        m = re.search(r"dil::(\w+)", name)
        if not m:
            continue
        inst = re.sub(r"^void dil::", "", name)
        inst = re.sub(r"\(.*$", "", inst)
        seen.setdefault(m.group(1), []).append((inst, calls, gmin, gmax, wg, avg / 1e3))
    print(f"{'kernel instantiation':64s} {'launches':>8s} {'blocks min..max':>18s} {'wg':>5s} {'avg_us':>9s}")
    for k in sorted(decl, key=lambda k: (decl[k], k)):
        if k not in seen:
            continue
        print(f"-- {k}  ({decl[k]})")
        for inst, calls, gmin, gmax, wg, avg in sorted(seen[k]):
            print(f"   {inst[:61]:61s} {calls:8d} {gmin:8d}..{gmax:<8d} {wg:5d} {avg:9.2f}")
    missing = [k for k in sorted(decl) if k not in seen]
    print(f"\n{len(decl) - len(missing)} of {len(decl)} __global__ functions launched by the traced tests")
    print("never launched: " + (", ".join(f"{k} ({decl[k]})" for k in missing) if missing else "none"))


if __name__ == "__main__":
    main()
