#!/bin/bash
# scratch: one GPU visit
bash scripts/gpu_r04.sh r04end tests smoke bench
