"""dil_ntt_host by batch size: a pageable caller buffer (memcpy through the library's ring of page-locked slots, 1 / 3 copy threads) and a
buffer the caller page-locked itself (DMA in place: round-robin / one stream per direction)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from dilithium_amd import api
from oracle.oracle import splitmix64_polys
api.init(0)
def med(f, reps=9):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
for size in (256, 1024, 2048, 4096, 4097, 8192, 12000, 16384, 32768, 65536, 131072):
    y = splitmix64_polys(size, seed=4); y0 = y.copy()
    pin = torch.empty((size, 256), dtype=torch.int32).pin_memory(); yp = pin.numpy(); yp[:] = y
    row = []
    for ct in (1, 3, 5, 8):
        api.set_option("host_copy_threads", ct)
        y[:] = y0; api.ntt(y); api.invntt(y); assert (y == y0).all()
        t = med(lambda: api.ntt(y)); row.append(f"pageable copy_threads={ct}: {t*1e3:6.3f} ms {size/t/1e6:5.1f} M/s")
    api.set_option("host_copy_threads", 3)
    for dp in (0, 1):
        api.set_option("host_duplex", dp)
        t = med(lambda: api.ntt(yp)); row.append(f"locked duplex={dp}: {t*1e3:6.3f} ms {size/t/1e6:5.1f} M/s")
    print(f"batch {size:6d}: " + "  ".join(row), flush=True)
