#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
echo "--- per-GPU rate of the strong-scaling form: 64 recordings over N GPUs = 64 / N on this one"
for n in 64 32 16 8; do python tools/kbench.py --precision fp32-split --batch $n --iters 80 --tag split_batch$n | cut -c1-420; done
for n in 32 16 8; do python tools/kbench.py --precision fp64 --batch $n --iters 60 --tag f64_batch$n | cut -c1-420; done
timeout 900 python -m pytest tests/test_gpu_configs.py -q -k "c5_all_nine" 2>&1 | tail -3
