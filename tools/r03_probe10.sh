#!/bin/bash
export VBX_AMD_NO_REBUILD=1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "two_level or long_recordings or c5 or c3" 2>&1 | tail -8
for g2 in 1 0; do
for cfg in "1 200000 50 fp32" "1 200000 50 fp64" "1 50000 30 fp32" "1 100000 30 fp32"; do
  set -- $cfg
  VBX_AMD_SCAN_GROUP2=$g2 timeout 300 python tools/kbench.py --batch $1 --T $2 --S $3 --precision $4 --iters 20 --tag "g2_${g2}_b$1_T$2_S$3_$4" 2>&1 | tail -1 | cut -c1-330
done
VBX_AMD_SCAN_GROUP2=$g2 timeout 300 python tools/kbench.py --sweep shared --T 200000 --S 50 --precision fp32 --iters 12 --tag "g2_${g2}_c5" 2>&1 | tail -1 | cut -c1-330
done
