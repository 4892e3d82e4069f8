#!/bin/bash
# round 5, first GPU call: the -m gpu suite on the changed library, the driver's bench command (is the last line parseable?),
# then A/B measurements that decide what to build next
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
out=gpurun_out
( time timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 ) 2>&1 | tail -14
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $out/r05_a_bench.out 2> $out/r05_a_bench.err ) 2>&1 | grep real
tail -n 1 $out/r05_a_bench.out | wc -c
tail -n 1 $out/r05_a_bench.out
echo "--- C5 sweep: half-tile re-runs on (library's choice) / off"
for st in 0 2; do VBX_AMD_SPLIT_TILES=$st python tools/kbench.py --sweep shared --T 200000 --S 50 --precision fp32-split --iters 10 --tag c5_split_tiles$st; done
VBX_AMD_SPLIT_TILES=2 python tools/kbench.py --sweep shared --T 200000 --S 50 --precision fp64 --iters 6 --tag c5_f64_tiles2
echo "--- headline: stream groups"
for s in 2 3 4; do VBX_AMD_STREAMS=$s python tools/kbench.py --precision fp32-split --tag split_streams$s; done
for s in 3 4; do VBX_AMD_STREAMS=$s python tools/kbench.py --precision fp64 --tag f64_streams$s; done
