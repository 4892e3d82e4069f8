#!/bin/bash
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
: > $out/r03_sweep2.jsonl
for cfg in "2 0" "1 0" "0 0" "2 2"; do
  set -- $cfg
  VBX_AMD_FUSE=$1 VBX_AMD_SPLIT_TILES=$2 timeout 300 python tools/kbench.py --sweep shared --T 200000 --S 50 --precision fp32 --iters 12 --tag "c5_shared_fuse$1_split$2" >> $out/r03_sweep2.jsonl 2>> $out/r03_sweep2.err
done
VBX_AMD_FUSE=1 timeout 300 python tools/kbench.py --sweep shared --T 200000 --S 50 --precision fp64 --iters 12 --tag "c5_shared_fuse1_fp64" >> $out/r03_sweep2.jsonl 2>> $out/r03_sweep2.err
for cfg in "1 0" "2 2"; do
  set -- $cfg
  VBX_AMD_FUSE=$1 VBX_AMD_SPLIT_TILES=$2 timeout 300 python tools/kbench.py --batch 64 --precision fp64 --iters 12 --tag "hl_fp64_fuse$1_split$2" >> $out/r03_sweep2.jsonl 2>> $out/r03_sweep2.err
done
cat $out/r03_sweep2.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ahc.py -x -q -m gpu -k "sweep or shared or score_matrix or c5" 2>&1 | tail -5
