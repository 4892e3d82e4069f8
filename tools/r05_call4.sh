#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
echo "--- C5 sweep, streams x fin block size"
for p in fp32-split fp64; do for st in 1 3; do for ft in 256 1024; do
  VBX_AMD_FIN_THREADS=$ft VBX_AMD_SWEEP_STREAMS=$st python tools/kbench.py --sweep shared --T 200000 --S 50 --precision $p --iters 8 --tag c5_${p}_streams${st}_fin$ft | cut -c1-420
done; done; done
echo "--- headline, fin block size"
for ft in 256 1024; do VBX_AMD_FIN_THREADS=$ft python tools/kbench.py --precision fp32-split --tag split_fin$ft | cut -c1-420; done
echo "--- N-rank bench path on one GPU"
timeout 600 python -m pytest tests/test_gpu_multirank.py -q -x -k "bench" 2>&1 | tail -5
