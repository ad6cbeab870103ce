#!/usr/bin/env python3
"""A/B builds of the library: scripts/build_variant.py <name> [-DFLAG=..]... -> scripts/bin/libdil256_<name>.so
(git-ignored; select at run time with DIL_LIB_PATH=scripts/bin/libdil256_<name>.so).  Variant builds -- and only they -- see
csrc/variants.hpp, where the ablation switches (-DDIL_ABL_NONTT, -DDIL_ABL_NOALOAD, ...) are defined."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dilithium_amd import _build  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "scripts", "bin", f"libdil256_{name}.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
cmd = ["hipcc"] + _build.FLAGS + ["-DDIL_VARIANT_BUILD"] + flags + [os.path.join(_build.CSRC, s) for s in _build.SOURCES] + ["-o", out]
subprocess.check_call(cmd)
print(out)
