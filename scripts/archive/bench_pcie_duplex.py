#!/usr/bin/env python3
"""Is the host link full duplex from one process?  H2D alone, D2H alone and both at once (two streams, page-locked buffers), by copy size;
then the same through N chunks per direction on 1..4 streams per direction -- the pattern of capi.hip's host pipelines."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

MiB = 1 << 20
total = 256 * MiB
hup = torch.empty(total, dtype=torch.uint8).pin_memory()
hdn = torch.empty(total, dtype=torch.uint8).pin_memory()
dup = torch.empty(total, dtype=torch.uint8, device="cuda")
ddn = torch.empty(total, dtype=torch.uint8, device="cuda")
streams = [torch.cuda.Stream() for _ in range(8)]


def run(chunk, ns, up=True, dn=True, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        k = 0
        for off in range(0, total, chunk):
            if up:
                with torch.cuda.stream(streams[k % ns]):
                    dup[off:off + chunk].copy_(hup[off:off + chunk], non_blocking=True)
            if dn:
                with torch.cuda.stream(streams[4 + k % ns]):
                    hdn[off:off + chunk].copy_(ddn[off:off + chunk], non_blocking=True)
            k += 1
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


print(f"256 MiB each way, page-locked host buffers; GB/s per direction")
for chunk in (256 * MiB, 64 * MiB, 16 * MiB, 4 * MiB, 1 * MiB, 256 * 1024):
    for ns in (1, 2, 4):
        tu, td, tb = run(chunk, ns, True, False), run(chunk, ns, False, True), run(chunk, ns, True, True)
        print(f"chunk {chunk // 1024:7d} KiB, {ns} stream(s) per direction: up alone {total / tu / 1e9:5.1f}  down alone {total / td / 1e9:5.1f}  "
              f"both at once {total / tb / 1e9:5.1f} each way ({2 * total / tb / 1e9:5.1f} in all)", flush=True)
