#!/bin/bash
# kernel timeline of one dil_sign_dev call: gpurun -- bash scripts/gpu_trace_sign.sh <tag> <level> <batch> <shared>
TAG=${1:-tr}; LV=${2:-3}; N=${3:-8192}; SH=${4:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/${TAG}_trace -o t -- python $GRAFT_REPO_ROOT/scripts/trace_sign.py $LV $N $SH > $OUT/${TAG}_trace.log 2>&1
python $GRAFT_REPO_ROOT/scripts/rocpd_timeline.py $OUT/${TAG}_trace/t_results.db 60 $OUT/${TAG}_sign_timeline_L${LV}_${N}_sh${SH}.txt
