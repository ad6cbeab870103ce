"""The two randomised soaks (scripts/fuzz_parity.py, scripts/fuzz_scheme.py), time-bounded, inside `-m gpu` so that the driver's GPU test
run executes them: random level / batch size / key mode / kernel shape of the fused pipelines against the oracle on every output, and
random keygen / sign_msg / verify_msg cases under random option settings with sampled items recomputed by the host KAT harness byte for
byte.  A fresh seed per run would make failures unreproducible: the seed is fixed, the budget is what bounds the walk."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("script,seconds,seed", [("fuzz_parity.py", 30, 41), ("fuzz_scheme.py", 30, 42)])
def test_bounded_fuzz_soak(gpu, script, seconds, seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(seconds), str(seed)], capture_output=True, text=True,
                         timeout=seconds + 240, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    last = [ln for ln in out.stdout.splitlines() if ln.startswith("fuzz_")][-1]
    assert "random cases" in last and int(last.split(":")[1].split()[0]) >= 5, last
    print(last)
