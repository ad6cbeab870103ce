"""One rank of a multi-process run of the sharded path (launched by tests/test_multi_gpu.py through torch.distributed.run):
slices a seeded batch with sharding.run_sharded, computes the level-5 sign inner loop (phase 1 + 2) and a forward NTT on its
slice -- with the HIP kernels on cuda:LOCAL_RANK (--compute hip, backend nccl = RCCL; or gloo with the ranks sharing a GPU when
DIL_DIST_BACKEND=gloo: the one-GPU rehearsal of the same code) or with the oracle on the CPU
(--compute oracle, backend gloo: exercises this script itself where there is no GPU) -- gathers the (z, h, flag) slabs and the
transformed polynomials, and rank 0 compares EVERYTHING with the oracle's unsharded result.  Exit code 0 = identical."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dilithium_amd import sharding  # noqa: E402
from oracle import dilithium_kat as dk  # noqa: E402
from oracle.oracle import Oracle, splitmix64_polys, Q  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--compute", choices=["hip", "oracle"], required=True)
    ap.add_argument("--items", type=int, default=4099)
    ap.add_argument("--level", type=int, default=5)
    a = ap.parse_args()
    hip = a.compute == "hip"
    # hip: nccl (= RCCL), or -- DIL_DIST_BACKEND=gloo -- a rehearsal in which the ranks share the GPUs that are there
    rank, world, local = sharding.init_distributed(None if hip else "gloo")
    o = Oracle()
    level, n = a.level, a.items
    p = dk.PARAMS[level]
    K, L = p.K, p.L
    rng = np.random.default_rng(9)                      # the same inputs on every rank
    A = splitmix64_polys(K * L, seed=8).reshape(1, K, L, 256)
    y = np.mod(rng.integers(-(p.gamma1 - 1), p.gamma1 + 1, (n, L, 256)), Q).astype(np.int32)
    c = np.zeros((n, 256), np.int32)
    c[:, ::5] = 1
    c[:, 1::9] = Q - 1
    s1h = o.ntt(np.mod(rng.integers(-p.eta, p.eta + 1, (1, L, 256)), Q).astype(np.int32))
    s2h = o.ntt(np.mod(rng.integers(-p.eta, p.eta + 1, (1, K, 256)), Q).astype(np.int32))
    t0h = o.ntt(np.mod(rng.integers(-4095, 4097, (1, K, 256)), Q).astype(np.int32))
    polys = splitmix64_polys(n, seed=5)
    if hip:
        from dilithium_amd import api
        dev = sharding.local_device(local)
        torch.cuda.set_device(dev)
        api.init(dev)
        d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()  # noqa: E731
        dA, ds1, ds2, dt0 = d(A), d(s1h), d(s2h), d(t0h)

        def sign_fn(yy, cc):
            w1, w0 = api.sign_phase1(dA, yy.contiguous(), level, shared_key=True)
            return api.sign_phase2(cc.contiguous(), yy.contiguous(), w0, w1, ds1, ds2, dt0, level, shared_key=True)

        def ntt_fn(x):
            return api.ntt(x.contiguous().clone())
    else:
        d = lambda x: torch.from_numpy(np.ascontiguousarray(x))  # noqa: E731

        def sign_fn(yy, cc):
            w1, w0 = o.sign_phase1(level, A, yy.numpy())
            z, h, f = o.sign_phase2(level, cc.numpy(), yy.numpy(), w0, w1, s1h, s2h, t0h)
            return torch.from_numpy(z), torch.from_numpy(h), torch.from_numpy(f)

        def ntt_fn(x):
            return torch.from_numpy(o.ntt(x.numpy()))
    z, h, f = sharding.run_sharded(sign_fn, n, d(y), d(c))
    t = sharding.run_sharded(ntt_fn, n, d(polys))
    sharding.barrier()
    rc = 0
    if rank == 0:
        ow1, ow0 = o.sign_phase1(level, A, y)
        oz, oh, of = o.sign_phase2(level, c, y, ow0, ow1, s1h, s2h, t0h)
        ok = (f.cpu().numpy() == of).all() and (z.cpu().numpy() == oz).all() and (h.cpu().numpy() == oh).all() and \
            (t.cpu().numpy() == o.ntt(polys)).all() and z.shape[0] == n
        print(f"rank0: world {world} compute {a.compute} items {n}: {'IDENTICAL' if ok else 'MISMATCH'}", flush=True)
        rc = 0 if ok else 1
    sharding.barrier()
    torch.distributed.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
