#!/usr/bin/env python3
"""batch-of-one host calls: the resident mailbox wave against a launch per call (include/dil256.h "HOST MAILBOX").
usage: bench_mailbox.py [calls]   (ctypes adds ~1 us per call to both columns)"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import dilithium_amd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = dilithium_amd.load()
assert L.dil_init(0) == 0
a = np.arange(256, dtype=np.int32)
b = np.arange(256, dtype=np.int32)[::-1].copy()
p, q = a.ctypes.data_as(C.POINTER(C.c_int32)), b.ctypes.data_as(C.POINTER(C.c_int32))
ops = {"ntt": lambda: L.dil_ntt_host(p, 1), "invntt": lambda: L.dil_invntt_host(p, 1), "pointwise": lambda: L.dil_pointwise_host(p, p, q, 1),
       "bram_fwdntt(AFTER_INVNTT)": lambda: L.dil_bram_fwdntt_host(p, 1, 2), "bram_mul(AFTER_NTT)": lambda: L.dil_bram_mul_host(p, q, 1, 1)}
for name, fn in ops.items():
    row = []
    for mode in (1, 0):
        L.dil_set_option(b"host_mailbox", mode)
        reps = n if mode else max(200, n // 10)
        for _ in range(50):
            fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            rc = fn()
        dt = time.perf_counter() - t0
        assert rc == 0
        row.append(dt / reps * 1e6)
    print(f"{name:28s} mailbox {row[0]:6.2f} us/call   launch path {row[1]:6.2f} us/call   ({row[1] / row[0]:.1f} x)")
c, l_, al = C.c_uint64(), C.c_uint64(), C.c_int()
L.dil_mailbox_stats(C.byref(c), C.byref(l_), C.byref(al))
print(f"mailbox: {c.value} calls served by {l_.value} launches of the resident wave")
