#!/bin/bash
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) 2>&1 | tail -20
for cfg in "1 10000 30 fp32" "1 10000 10 fp32" "1 50000 30 fp32" "64 10000 30 fp32" "64 10000 30 fp64"; do
  set -- $cfg
  timeout 300 python tools/kbench.py --batch $1 --T $2 --S $3 --precision $4 --iters 20 --tag "fin_b$1_T$2_S$3_$4" 2>&1 | tail -1
done
