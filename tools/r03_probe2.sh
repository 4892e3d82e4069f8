#!/bin/bash
# second round-3 measurement: atomic-accumulation probe, the shared-rho sweep against private copies (times + tests)
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/atomic_probe tools/atomic_probe.hip && /tmp/atomic_probe > $out/r03_atomic_probe.txt 2>&1
cat $out/r03_atomic_probe.txt
: > $out/r03_sweep.jsonl
for cfg in "private fp32" "shared fp32" "private fp64" "shared fp64"; do
  set -- $cfg
  timeout 300 python tools/kbench.py --sweep $1 --T 200000 --S 50 --precision $2 --iters 20 --tag "c5_$1_$2" >> $out/r03_sweep.jsonl 2>> $out/r03_sweep.err
done
cat $out/r03_sweep.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_ahc.py -x -q -m gpu -k "sweep or shared or score_matrix or c5" 2>&1 | tail -15
