#!/usr/bin/env python3
"""kernel resource usage of a .hip file from hipcc's -Rpass-analysis remarks: VGPRs, AGPRs, spills, LDS, occupancy
usage: scripts/kres.py dilithium_amd/csrc/pipelines.hip [substring] [-DFLAG ...]"""
import re
import subprocess
import sys

f = sys.argv[1]
extra = [x for x in sys.argv[2:] if x.startswith("-D")]
rest = [x for x in sys.argv[2:] if not x.startswith("-D")]
filt = rest[0] if rest else ""
out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only",
                      "-Rpass-analysis=kernel-resource-usage", *extra, f, "-o", "/tmp/kres.o"], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|[A-Za-z ]+?)(?: \[[^\]]*\])?: +(\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>5s} {'SGPR':>5s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    if filt in r["name"]:
        print(f"{r['name'][-58:]:58s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>5s} "
              f"{r.get('TotalSGPRs', '?'):>5s} {r.get('LDS Size', '?'):>7s} {r.get('Occupancy', '?'):>4s}")
