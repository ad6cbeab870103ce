#!/usr/bin/env python3
"""Print the kernel timeline (start offset, duration, gap to previous) of the LAST `count` dispatches in a rocprofv3 rocpd DB.
usage: rocpd_timeline.py <results.db> <count> [out.txt]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2])
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start desc limit ?", (n,)).fetchall()[::-1]
t0 = rows[0][1]
prev_end = t0
lines = [f"{'t_us':>10s} {'dur_us':>9s} {'gap_us':>8s} {'grid':>9s}  kernel"]
for name, st, en, gx, wx in rows:
    short = name.replace("void dil::", "").replace("dil::", "").split("(")[0][:60]
    lines.append(f"{(st - t0) / 1e3:10.1f} {(en - st) / 1e3:9.1f} {(st - prev_end) / 1e3:8.1f} {gx:9d}  {short}")
    prev_end = en
lines.append(f"span {(rows[-1][2] - t0) / 1e3:.1f} us, busy {sum(r[2] - r[1] for r in rows) / 1e3:.1f} us")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write(out + "\n")
