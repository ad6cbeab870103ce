"""Whole-call latencies and rates with the cooperative Keccak forms on (option coop_max at its default) and off (0), interleaved:
keygen / sign / verify at batch 1, 64, 1024 (wall time per call, the way bench.py's `latency` block measures it) and the signing / verification
rates at 8192 (HIP events around back-to-back calls).  Outputs are compared between the two settings.
    python scripts/bench_coop.py [level]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dilithium_amd import api  # noqa: E402
from dilithium_amd import lib as dlib  # noqa: E402


def main():
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    api.init(0)
    L = dlib.load()
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cuda").manual_seed(5)
    u8 = lambda *sh: torch.randint(0, 256, sh, dtype=torch.uint8, device="cuda", generator=g)  # noqa: E731
    NB = 8192
    seed, mu = u8(NB, 32), u8(NB, 64)
    pk, sk = api.keygen(seed, level)
    sig, att = api.sign(sk, mu, level, shared_sk=False)
    default = api.get_option("coop_max")
    pkb, skb, sgb = api.pk_bytes(level), api.sk_bytes(level), api.sig_bytes(level)
    o_pk = torch.empty((NB, pkb), dtype=torch.uint8, device="cuda")
    o_sk = torch.empty((NB, skb), dtype=torch.uint8, device="cuda")
    o_sig = torch.empty((NB, sgb), dtype=torch.uint8, device="cuda")
    o_att = torch.empty((NB,), dtype=torch.int32, device="cuda")
    o_vd = torch.empty((NB,), dtype=torch.int32, device="cuda")
    blob = u8(NB * 64)
    offs = (torch.arange(NB, device="cuda", dtype=torch.int64) * 64).contiguous()
    lens = torch.full((NB,), 64, dtype=torch.int32, device="cuda")

    calls = {
        "keygen": lambda n: L.dil_keygen_dev(P(o_pk), P(o_sk), P(seed), level, n, stream),
        "sign (one key)": lambda n: L.dil_sign_dev(P(o_sig), P(o_att), P(sk), P(mu), level, n, 1, 512, stream),
        "sign (key/item)": lambda n: L.dil_sign_dev(P(o_sig), P(o_att), P(sk), P(mu), level, n, 0, 512, stream),
        "verify (key/item)": lambda n: L.dil_verify_sig_dev(P(o_vd), P(pk), P(sig), P(mu), level, n, 0, stream),
        "verify (one key)": lambda n: L.dil_verify_sig_dev(P(o_vd), P(pk), P(sig), P(mu), level, n, 1, stream),
        "sign_msg 64 B (one key)": lambda n: L.dil_sign_msg_dev(P(o_sig), P(o_att), P(sk), P(blob), blob.numel(), P(offs), P(lens), level, n, 1, 512, stream),
    }

    def wall_us(fn, n, reps):
        for _ in range(3):
            dlib.check(fn(n))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(n)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e6

    print(f"level {level}: microseconds per call (wall, back-to-back calls), coop_max = {default} vs 0")
    for name, fn in calls.items():
        for n in (1, 64, 1024, 8192):
            reps = 40 if n <= 1024 else 12
            res = {}
            for rnd in range(3):
                for setting in (default, 0):
                    api.set_option("coop_max", setting)
                    res.setdefault(setting, []).append(wall_us(fn, n, reps))
            a, b = sorted(res[default])[1], sorted(res[0])[1]
            print(f"  {name:26s} n={n:5d}: coop {a:9.1f} us   lane/two-lane {b:9.1f} us   x{b / a:5.2f}" +
                  (f"   {n / a:7.2f} vs {n / b:7.2f} M/s" if n >= 1024 else ""))
    # identical outputs under the two settings (sign: the first 1024)
    outs = {}
    for setting in (default, 0):
        api.set_option("coop_max", setting)
        s2, a2 = api.sign(sk[:1024].contiguous(), mu[:1024].contiguous(), level, shared_sk=False)
        v2 = api.verify_sig(pk[:1024].contiguous(), s2, mu[:1024].contiguous(), level)
        outs[setting] = (s2, a2, v2)
    same = all(torch.equal(x, y) for x, y in zip(outs[default], outs[0]))
    print("signatures / attempt counts / verdicts identical under both settings:", same, " all accepted:", int(outs[default][2].abs().sum()) == 0)
    api.set_option("coop_max", default)


if __name__ == "__main__":
    main()
