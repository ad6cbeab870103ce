"""The memcpy pool of the host-pointer entry points (dilithium_amd/csrc/copy_pool.hpp: calling thread + parked pool threads, pure C++, no
HIP) under ThreadSanitizer, without a GPU: three caller threads, random sizes / offsets / thread counts, every byte compared, nothing written
outside the destination.  The reference's contract for these calls is the caller's buffer and nothing else (reference_code/ref_ntt.h:30-36)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_copy_pool_under_sanitizers(tmp_path, san):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    exe = str(tmp_path / "test_copy_pool")
    build = subprocess.run([cxx, "-O1", "-g", "-std=c++17", f"-fsanitize={san}", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_copy_pool.cpp"),
                            "-o", exe], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr.lower():
        pytest.skip("sanitizer runtime not installed: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe, "25"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert "mismatches 0" in run.stdout and "WARNING: ThreadSanitizer" not in run.stderr
