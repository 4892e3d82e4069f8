#!/bin/bash
# Round-3 opening measurement (one gpurun call): per-kernel times of every BASELINE.json configuration in both
# precisions (tools/kbench.py), then the rocprofv3 passes of the fp64 headline batch (tools/profile_bench.sh).
# usage (GPU box): bash tools/r03_probe.sh
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
: > $out/r03_kbench.jsonl
for cfg in "64 10000 30 fp32" "64 10000 30 fp64" "1 10000 10 fp32" "1 10000 10 fp64" "1 50000 30 fp32" "1 50000 30 fp64" \
           "9 200000 50 fp32" "9 200000 50 fp64" "1 200000 50 fp32" "1 10000 30 fp32"; do
  set -- $cfg
  echo "== batch=$1 T=$2 S=$3 $4" >> $out/r03_kbench.err
  timeout 300 python tools/kbench.py --batch $1 --T $2 --S $3 --precision $4 --iters 20 --tag "b$1_T$2_S$3_$4" >> $out/r03_kbench.jsonl 2>> $out/r03_kbench.err
done
cat $out/r03_kbench.jsonl
SQ_EXTRA="SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS" bash tools/profile_bench.sh r03_f64_s1 --precision fp64 --streams 1 > $out/r03_prof_f64.log 2>&1
tail -30 $out/r03_prof_f64.log
