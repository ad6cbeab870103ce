#!/usr/bin/env python3
"""Experiment (round-6 review item 4): the UPPER BOUND of a host-free signing loop.

A loop whose rounds size themselves from a count in device memory could at best run as if the host had known every round's pending count
in advance: no read-back, no wake-up, every launch queued behind the previous one, grids and kernel forms exactly as the host-sized loop
picks them.  This script builds that bound as a VARIANT library (never shipped): a copy of csrc/scheme.hip in which sign_core takes the
pending counts of the rounds from the environment (DIL_SIGN_REPLAY="n1,n2,...": signing is deterministic, so a first ordinary call of the
same inputs gives them) instead of waiting for them, and checks the device's count once, after the last round (option sign_wake = 1 assumed: the default).

    python scripts/sign_replay/apply.py            ->  scripts/bin/libdil256_replay.so   (other objects: dilithium_amd/build/*.o)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "dilithium_amd", "csrc")
src = open(os.path.join(CSRC, "scheme.hip")).read()

OLD_SYNC = """        if (wake_flag) {
            if ((rc = await_round_count(host_counts, counts, lose_post ? seq ^ 0x80000000u : seq, s))) return rc;
        } else {
            DIL_TRY(hipEventSynchronize(counted.ev));
        }
        n = (size_t)host_counts[0];
"""
NEW_SYNC = """        if (round_no < replay.size()) {
            n = replay[round_no];
            if (round_no + 1 == replay.size()) {          // the one wait a host-free loop keeps: the count after its last queued round
                if ((rc = await_round_count(host_counts, counts, seq, s))) return rc;
                if ((size_t)host_counts[0] != n) return (int)hipErrorAssert;          // the replayed schedule was not this input's
            }
        } else {
            if ((rc = await_round_count(host_counts, counts, seq, s))) return rc;
            n = (size_t)host_counts[0];
        }
        round_no++;
"""
OLD_LOOP = """    int a0 = 0;                                          // attempts every pending item has already failed
"""
NEW_LOOP = OLD_LOOP + """    std::vector<size_t> replay;
    size_t round_no = 0;
    if (const char* e = getenv("DIL_SIGN_REPLAY"))
        for (const char* q = e; *q;) {
            char* end;
            replay.push_back((size_t)strtoull(q, &end, 10));
            q = *end == ',' ? end + 1 : end;
            if (end == q && *q) break;
        }
"""
for old, new in ((OLD_SYNC, NEW_SYNC), (OLD_LOOP, NEW_LOOP)):
    assert src.count(old) == 1, old
    src = src.replace(old, new)
src = src.replace('#include "capi_internal.hpp"', '#include <vector>\n#include <cstdlib>\n#include "capi_internal.hpp"', 1)

out_dir = os.path.join(ROOT, "scripts", "bin")
os.makedirs(out_dir, exist_ok=True)
var = os.path.join(CSRC, "_scheme_replay.hip")          # beside the original: its #include "..." lines resolve as they are
open(var, "w").write(src)
try:
    obj = os.path.join(out_dir, "scheme_replay.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-pthread", "-c", var, "-o", obj])
finally:
    os.remove(var)
objs = [os.path.join(ROOT, "dilithium_amd", "build", f) for f in sorted(os.listdir(os.path.join(ROOT, "dilithium_amd", "build")))
        if f.endswith(".o") and f != "scheme.o"]
lib = os.path.join(out_dir, "libdil256_replay.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + [obj, "-o", lib])
print(lib)
