cd $GRAFT_REPO_ROOT
bash scripts/gpu_r04.sh r04y tests cover smoke bench prof pmc signpmc > gpurun_out/r04y_round.log 2>&1
tail -5 gpurun_out/r04y_round.log | cut -c1-300
