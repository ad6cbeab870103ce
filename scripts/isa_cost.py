#!/usr/bin/env python3
"""Estimate VALU issue cycles of a kernel's main loop from the gfx950 ISA, weighting each
instruction with the issue cost measured by scripts/ubench*.hip (profiles/r01_ubench_valu_rates.txt).
usage: isa_cost.py <file.s> <kernel-name-substring>"""
import collections
import re
import sys

FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_ashrrev_i32", "v_lshrrev_b32", "v_and_b32", "v_xor_b32",
        "v_or_b32", "v_mov_b32", "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32"}


def cost(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.startswith("s_") or op.startswith("global_") or op.startswith("ds_") or op.startswith("buffer_"):
        return 0.0
    if "permlane" in base:
        return 8.2
    if base.startswith("v_mad_u64") or base.startswith("v_mad_i64"):
        return 5.2
    if base == "v_cndmask_b32" and op.endswith("e32"):
        return 22.0
    if op.endswith("_dpp"):
        return 4.4
    if base in FAST:
        return 2.5
    return 4.4


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    for f in re.split(r"\n(?=_Z[^\n]*:\s*; @)", s)[1:]:
        name = f.split(":")[0]
        if key not in name:
            continue
        lines = f.split("\n")
        # innermost loop = between the last loop-header label and its back-branch
        labels = [i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)]
        best = None
        for i in labels:
            lab = lines[i].split(":")[0]
            for j in range(i + 1, len(lines)):
                if re.search(r"s_cbranch_\w+\s+" + re.escape(lab) + r"\b", lines[j]):
                    if best is None or (j - i) > (best[1] - best[0]):
                        best = (i, j)
                    break
        if best is None:
            print(name, "no loop found")
            continue
        ops = [l.strip().split()[0] for l in lines[best[0]:best[1] + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = collections.Counter(ops)
        tot = sum(cost(o) * n for o, n in c.items())
        valu = sum(n for o, n in c.items() if o.startswith("v_"))
        print(f"{name[:70]}: loop {len(ops)} instr, {valu} VALU, est. {tot:.0f} issue cycles/iter")
        for o, n in c.most_common(14):
            print(f"    {n:4d} {o:28s} {cost(o) * n:7.1f}")


if __name__ == "__main__":
    main()
