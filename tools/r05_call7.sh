#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
echo "--- stream priorities: headline (split, exact, fp64), 3 and 4 streams"
for pr in 0 1; do for s in 3 4; do
  VBX_AMD_STREAM_PRIO=$pr VBX_AMD_STREAMS=$s python tools/kbench.py --precision fp32-split --iters 100 --tag split_prio${pr}_streams$s | cut -c150-420
done; done
for pr in 0 1; do
  VBX_AMD_STREAM_PRIO=$pr python tools/kbench.py --precision fp32 --iters 100 --tag exact_prio${pr} | cut -c150-420
  VBX_AMD_STREAM_PRIO=$pr python tools/kbench.py --precision fp64 --iters 60 --tag f64_prio${pr} | cut -c150-420
  VBX_AMD_STREAM_PRIO=$pr python tools/kbench.py --sweep shared --T 200000 --S 50 --precision fp32-split --iters 8 --tag c5_prio${pr} | cut -c150-460
  VBX_AMD_STREAM_PRIO=$pr python tools/kbench.py --precision fp32-split --batch 32 --iters 100 --tag split32_prio${pr} | cut -c150-420
done
