#!/usr/bin/env python3
"""Read a dump of scripts/hiptrace.c beside the fault address ROCr printed ("Memory access fault by GPU ... on address 0x...") and say what the
process did with that page: every copy / registration / madvise whose range covers it (or comes within --near bytes), which thunk registrations
(madvise DONTFORK without a later DOFORK of the same range) were alive when the process died, and the last calls of every thread.
usage: hiptrace_report.py <hiptrace_PID.txt> <fault address, hex> [--near BYTES]"""
import sys


def parse(path):
    lines = open(path, errors="replace").read().split("\n")
    i0 = next(i for i, ln in enumerate(lines) if ln.startswith("---- of the last") or ln.startswith("---- last"))
    maps = [ln for ln in lines[:i0] if "-" in ln.split(" ")[0] and ln[:1] in "0123456789abcdef"]
    recs = []
    for ln in lines[i0 + 1:]:
        p = ln.split(" ")
        if len(p) < 8:
            continue
        try:
            k = next(j for j, x in enumerate(p[2:]) if x.startswith("0x") or x == "(nil)") + 2
            a, b, n, st, rc = p[k:k + 5]
            recs.append(dict(t=float(p[0]), tid=int(p[1]), what=" ".join(p[2:k]), a=0 if a == "(nil)" else int(a, 16),
                             b=0 if b == "(nil)" else int(b, 16), n=int(n), stream=st, rc=int(rc), sym=" ".join(p[k + 5:])))
        except (StopIteration, ValueError):
            continue
    return lines[:i0], maps, recs


def main():
    path, fault = sys.argv[1], int(sys.argv[2], 16)
    near = int(sys.argv[sys.argv.index("--near") + 1]) if "--near" in sys.argv else 0
    head, maps, recs = parse(path)
    tend = recs[-1]["t"]
    print(head[0])
    for ln in maps:
        lo, hi = (int(x, 16) for x in ln.split(" ")[0].split("-"))
        if lo <= fault < hi:
            print("fault address lies in:", ln.strip())

    def covers(r):
        if "Launch" in r["what"] or "Synchronize" in r["what"]:
            return False
        ptrs = [r["a"], r["b"]] if "Memcpy" in r["what"] else [r["a"]]
        return any(p and p - near <= fault < p + max(r["n"], 1) + near for p in ptrs)
    print(f"\n-- every recorded call whose range covers {fault:#x}" + (f" (+- {near})" if near else "") + ", oldest first (t relative to the last recorded call)")
    for r in recs:
        if covers(r):
            print(f"{r['t'] - tend:+12.6f}s tid {r['tid']:6d} {r['what']:26s} a={r['a']:#x} b={r['b']:#x} n={r['n']} stream={r['stream']} rc={r['rc']}")
    alive = {}
    for r in recs:
        if r["what"] == "madvise DONTFORK":
            alive[(r["a"], r["n"])] = r
        elif r["what"] == "madvise DOFORK":
            alive.pop((r["a"], r["n"]), None)
    print("\n-- thunk registrations (MADV_DONTFORK without a matching MADV_DOFORK) alive at the end that cover the fault page:")
    for (a, n), r in sorted(alive.items()):
        if a <= fault < a + n:
            print(f"{r['t'] - tend:+12.6f}s tid {r['tid']:6d} [{a:#x}, {a + n:#x}) {n} bytes")
    print(f"   ({len(alive)} registrations alive in all; {sum(r['what'] == 'madvise DONTFORK' for r in recs)} made, "
          f"{sum(r['what'] == 'madvise DOFORK' for r in recs)} released in the recorded window)")
    print("\n-- the last 25 recorded calls")
    for r in recs[-25:]:
        print(f"{r['t'] - tend:+12.6f}s tid {r['tid']:6d} {r['what']:26s} a={r['a']:#x} b={r['b']:#x} n={r['n']} stream={r['stream']} rc={r['rc']} {r['sym'][:50]}")


if __name__ == "__main__":
    main()
