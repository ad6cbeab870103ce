#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (what a caller of the reference's
host-buffer API sees).  Never the headline `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dilithium_amd import api
from oracle.oracle import splitmix64_polys

api.init(0)
n = 65536
a = splitmix64_polys(n, seed=3)
ref = None
for rep in range(4):
    x = a.copy()
    t0 = time.perf_counter(); api.ntt(x); t1 = time.perf_counter(); api.invntt(x); t2 = time.perf_counter()
    assert (x == a).all()
    print(f"pin={os.environ.get('DIL_HOST_PIN','0')} host ntt {n}: {(t1-t0)*1e3:7.2f} ms ({n/(t1-t0)/1e6:6.1f} M NTT/s, {2*n*1024/(t1-t0)/1e9:5.1f} GB/s both ways)  invntt {(t2-t1)*1e3:7.2f} ms")
