#!/bin/bash
# bench.py's multi-rank control flow (barriers, max-over-ranks, region counts, sharded configs[4], gathers) rehearsed on a ONE-GPU box:
# N ranks over gloo sharing GPU 0 (DIL_DIST_BACKEND=gloo; the numbers mean nothing, the point is that every rank reaches every
# collective the same number of times and rank 0 prints the line).   gpurun --timeout 1500 -- bash scripts/gpu_rehearse_ranks.sh [N]
N=${1:-2}
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
export DIL_DIST_BACKEND=gloo
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus $N --steps 20 --warmup 5 > $OUT/rehearse_${N}.log 2> $OUT/rehearse_${N}.err
echo "rc=$?" | tee -a $OUT/rehearse_${N}.log
grep '^{' $OUT/rehearse_${N}.log | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('n_gpus',d['n_gpus'],'value',d['value'],'regions',d['timing']['regions'])
s=d.get('secondary',{})
print('secondary keys',list(s))
print('configs4_sharded',json.dumps(s.get('configs4_sharded'))[:900])
"
tail -5 $OUT/rehearse_${N}.err
