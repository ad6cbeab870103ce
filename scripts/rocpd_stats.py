#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel trace [+ PMC]) into the text table committed
under profiles/.  usage: rocpd_stats.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name, grid_x order by 6 desc").fetchall()   # one row per (kernel, launch size)
    tot = sum(r[5] for r in rows) or 1
    lines = [f"{'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'%':>6s} "
             f"{'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'grid':>8s} {'wg':>5s}"]
    for r in rows:
        lines.append(f"{r[0][:72]:72s} {r[1]:6d} {r[2] / 1e3:9.2f} {r[3] / 1e3:9.2f} {r[4] / 1e3:9.2f} {r[5] / 1e6:9.3f} "
                     f"{100 * r[5] / tot:6.2f} {r[6]:5d} {r[7]:5d} {r[8]:6d} {r[9]:8d} {r[10]:5d}")
    try:
        pm = cur.execute(
            "select name, counter_name, avg(v), count(*) from (select substr(name, 1, 60) as name, counter_name, "
            "dispatch_id, sum(counter_value) as v from pmc_events group by dispatch_id, counter_name) "
            "group by name, counter_name").fetchall()
        if pm:
            lines.append("")
            lines.append("PMC (summed over SEs/XCCs, average per dispatch)")
            for name, cn, v, n in pm:
                if "dil::" in name or "--all" in sys.argv:
                    lines.append(f"{name:60s} {cn:28s} {v:18.1f}  (dispatches={n})")
    except Exception as e:  # noqa: BLE001
        lines.append(f"(no PMC table: {e})")
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
