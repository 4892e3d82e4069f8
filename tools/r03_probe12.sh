#!/bin/bash
export VBX_AMD_NO_REBUILD=1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -4
for lib in default nc1; do
  if [ $lib != default ]; then export VBX_AMD_LIB=$GRAFT_REPO_ROOT/vbx_amd/csrc/libvbx_hip_$lib.so; fi
  for cfg in "64 10000 30 fp32" "64 10000 30 fp64" "1 10000 30 fp32" "64 10000 50 fp32"; do
    set -- $cfg
    timeout 300 python tools/kbench.py --batch $1 --T $2 --S $3 --precision $4 --iters 30 --tag "${lib}_b$1_T$2_S$3_$4" 2>&1 | tail -1 | cut -c1-330
  done
  timeout 300 python tools/kbench.py --sweep shared --T 200000 --S 50 --precision fp32 --iters 12 --tag "${lib}_c5" 2>&1 | tail -1 | cut -c1-330
  timeout 300 python tools/kbench.py --sweep shared --T 200000 --S 50 --precision fp64 --iters 12 --tag "${lib}_c5_f64" 2>&1 | tail -1 | cut -c1-330
done
