"""Build libdil256.so (HIP kernels + C-ABI) for gfx950, in tree.

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libdil256.so")
OBJ_DIR = os.path.join(PKG, "build")       # git-ignored
# reference-identical C++ signatures (include/dil256_ref.hpp); DIL_REF_LIB_PATH: another build of it (scripts/san_check.sh)
REF_LIB_BUILT = os.path.join(PKG, "libdil256_ref.so")                      # what build_ref() writes
REF_LIB = os.environ.get("DIL_REF_LIB_PATH", REF_LIB_BUILT)              # what the tests load
SOURCES = ["kernels.hip", "pipelines.hip", "hash_kernels.hip", "coop_kernels.hip", "codec_kernels.hip", "wire_kernels.hip", "capi.hip", "scheme.hip", "multi_gpu.hip"]
HEADERS = ["capi_internal.hpp", "modarith.hpp", "ntt_core.hpp", "kernels.hpp", "device_common.hpp", "pipeline_common.hpp", "launch_util.hpp", "keccak.hpp", "keccak_coop.hpp", "coop_bodies.hpp", "wire_common.hpp", "sampler_bodies.hpp", "copy_pool.hpp", "ref_api.cpp", os.path.join("..", "..", "include", "dil256.h"),
           os.path.join("..", "..", "include", "dil256_ref.hpp")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-pthread"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _compile_one(hipcc: str, src: str, obj: str, verbose: bool) -> None:
    cmd = [hipcc] + [f for f in FLAGS if f != "-shared"] + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)


def build(force: bool = False, verbose: bool = False) -> str:
    """one object per translation unit (in parallel; only the stale ones unless `force`), then the link"""
    if not force and not _stale():
        if not os.path.exists(REF_LIB_BUILT):
            build_ref(verbose)
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libdil256.so")
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr_time = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))
    jobs, objs = [], []
    for src_name in SOURCES:
        src, obj = os.path.join(CSRC, src_name), os.path.join(OBJ_DIR, src_name.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append((src, obj))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as pool:
        for f in [pool.submit(_compile_one, hipcc, src, obj, verbose) for src, obj in jobs]:
            f.result()
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread"] + objs + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    build_ref(verbose)
    return LIB


def build_ref(verbose: bool = False) -> str:
    """libdil256_ref.so: host-only C++ (no device code), links against libdil256.so next to it"""
    cxx = shutil.which("g++") or shutil.which("hipcc")
    cmd = [cxx, "-O2", "-std=c++17", "-shared", "-fPIC", "-Wall", os.path.join(CSRC, "ref_api.cpp"),
           "-L" + PKG, "-ldil256", "-Wl,-rpath,$ORIGIN", "-o", REF_LIB_BUILT]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return REF_LIB_BUILT


if __name__ == "__main__":
    print(build(force=True, verbose=True))
