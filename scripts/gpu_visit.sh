#!/bin/bash
# scratch: one GPU visit
bash scripts/gpu_r04.sh r04zz tests cover smoke bench prof pmc signpmc
bash scripts/gpu_scale.sh 100
