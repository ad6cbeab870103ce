#!/usr/bin/env python3
"""What does a *_host call leave registered with the driver after it has returned?  Host-pointer transforms on heap arrays of several size
classes, then /proc/self/smaps: VMAs overlapping the caller's arrays that still carry VM_DONTCOPY (`dc`) -- the mark the thunk puts on every
host range it registers (a page-lock the runtime made for a pageable copy, or hipHostRegister) and clears when the registration goes.
usage: check_retained.py <tree root>      (the round-5 tree under _old/ against the current one: profiles/r06_retained_registrations.txt)"""
import ctypes
import os
import sys

root = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, root)
import numpy as np  # noqa: E402
from dilithium_amd import api  # noqa: E402
from oracle.oracle import splitmix64_polys  # noqa: E402


def registered(arr):
    lo, hi = arr.ctypes.data, arr.ctypes.data + arr.nbytes
    hits, cur = [], None
    for ln in open("/proc/self/smaps"):
        head = ln.split(" ", 1)[0]
        if "-" in head and ln[:1] in "0123456789abcdef":
            cur = tuple(int(x, 16) for x in head.split("-"))
        elif ln.startswith("VmFlags:") and cur and cur[0] < hi and cur[1] > lo and " dc" in ln:
            hits.append(cur)
    return hits


ctypes.CDLL("libc.so.6").mallopt(-3, 1 << 30)         # large arrays from the heap proper (as in a long-running process)
api.init(0)
print(f"library: {os.path.join(root, 'dilithium_amd', 'libdil256.so')}")
total = 0
for n in (300, 5000, 9000, 20000, 70000):
    a = splitmix64_polys(n, seed=n)
    x = a.copy()
    api.ntt(x)
    api.invntt(x)
    assert (x == a).all()
    left = registered(x)
    total += len(left)
    print(f"  dil_ntt_host + dil_invntt_host on {n:6d} polynomials ({x.nbytes >> 10:6d} KiB at {x.ctypes.data:#x}): "
          f"{len(left)} range(s) of the buffer still registered after the calls returned"
          + (": " + ", ".join(f"[{lo:#x}, {hi:#x})" for lo, hi in left[:4]) if left else ""))
print(f"  => {total} registered range(s) left behind")
