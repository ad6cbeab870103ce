#!/usr/bin/env python3
"""Timeline of the LAST burst of kernels in a rocprofv3 rocpd database: start offset, duration and the idle gap before each
launch -- where a multi-launch call (the signing loop's rounds) spends time that no kernel accounts for.
usage: rocpd_timeline.py <results.db> [gap_us_that_separates_calls = 300]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    sep = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 300e3
    rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    rows = [r for r in rows if "at::" not in r[0]]
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] > sep:
            cut = i
    burst = rows[cut:]
    t0 = burst[0][1]
    busy = 0
    gaps = 0
    prev_end = t0
    print(f"{'t_us':>8s} {'gap_us':>7s} {'dur_us':>8s} {'grid':>8s}  kernel")
    for name, st, en, gx, wx in burst:
        gap = st - prev_end
        print(f"{(st - t0) / 1e3:8.1f} {gap / 1e3:7.1f} {(en - st) / 1e3:8.1f} {gx:8d}  {name[:110]}")
        busy += en - st
        gaps += max(gap, 0)
        prev_end = max(prev_end, en)
    print(f"launches {len(burst)}  span {(prev_end - t0) / 1e3:.1f} us  kernels {busy / 1e3:.1f} us  idle between launches {gaps / 1e3:.1f} us")


if __name__ == "__main__":
    main()
