bash scripts/gpu_option_matrix.sh r04
