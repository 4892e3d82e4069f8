#!/bin/bash
# round 5, second GPU call: the suite with the referee and the cross-stream sweeps, then the C5 sweep on 1 / 2 / 3 streams
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
out=gpurun_out
( time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -25 ) 2>&1 | tail -32
cp $out/config_parity.json $out/r05_config_parity_b.json 2>/dev/null
echo "--- C5 sweep on 1 / 2 / 3 streams (one rho per stream)"
for p in fp32-split fp32 fp64; do for st in 1 2 3; do
  VBX_AMD_SWEEP_STREAMS=$st python tools/kbench.py --sweep shared --T 200000 --S 50 --precision $p --iters 8 --tag c5_${p}_streams$st | cut -c1-400
done; done
echo "--- a shorter sweep: T = 50 000, S = 30, nine points"
for st in 1 2 3; do VBX_AMD_SWEEP_STREAMS=$st python tools/kbench.py --sweep shared --T 50000 --S 30 --precision fp32-split --iters 12 --tag c3sweep_streams$st | cut -c1-400; done
