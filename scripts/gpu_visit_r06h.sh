#!/bin/bash
# Round-6 visit h: the driver's own command five times in sequence on one box (no GC fixtures, nothing preloaded), then smoke, bench, kernel stats, PMC.
cd $GRAFT_REPO_ROOT
STRESS_PLAIN=1 KEEP_GOING=1 bash scripts/stress_suite.sh r06h_plain ${1:-5}
bash scripts/gpu_r06.sh r06h smoke bench prof pmc verifyprof
