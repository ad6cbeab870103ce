cd $GRAFT_REPO_ROOT
bash scripts/gpu_r04.sh r04z tests cover smoke bench prof pmc signpmc > gpurun_out/r04z_round.log 2>&1
tail -3 gpurun_out/r04z_round.log | cut -c1-200
bash scripts/gpu_scale.sh 100 > gpurun_out/r04z_scale.log 2>&1; tail -1 gpurun_out/r04z_scale.log
