#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
echo "--- headline: stream groups with the flat walk forced (is the two-level walk of <= 16 recordings what hurts 4 streams?)"
for s in 3 4 5 6; do for g in 0 1; do
  VBX_AMD_STREAMS=$s VBX_AMD_SCAN_GROUP=$g python tools/kbench.py --precision fp32-split --iters 60 --tag split_streams${s}_group$g | cut -c1-420
done; done
for s in 3 4; do VBX_AMD_STREAMS=$s VBX_AMD_SCAN_GROUP=1 python tools/kbench.py --precision fp64 --iters 40 --tag f64_streams${s}_group1 | cut -c1-420; done
