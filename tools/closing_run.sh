#!/bin/bash
# The closing run of a round in ONE gpurun call:  tools/closing_run.sh <round tag, e.g. r04>
#   the -m gpu suite, smoke(), the round's profiles (tools/profiles.sh: they are stamped with the hash of the kernel sources
#   and copied into profiles/ on the box), THEN the two bench lines, which quote the profiles they find.
# What comes back: gpurun_out/profiles_<r>/* (copy into profiles/), gpurun_out/<r>_config_parity.json, gpurun_out/<r>_bench_*.json
r=${1:-r06}
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
out=gpurun_out
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) 2>&1 | tail -8
cp $out/config_parity.json $out/${r}_config_parity.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
( time bash tools/profiles.sh $r > $out/${r}_profiles.log 2>&1 ) 2>&1 | grep real
# (stdout of bench.py = the compact line the driver parses; the full record goes to --full-out; stderr repeats it)
( time python bench.py --full-out $out/${r}_bench_default.json > $out/${r}_bench_default_compact.json 2> /dev/null ) 2>&1 | grep real
( time python bench.py --steps 20 --warmup 5 --full-out $out/${r}_bench_driver_args.json > $out/${r}_bench_driver_args_compact.json 2> /dev/null ) 2>&1 | grep real
wc -c $out/${r}_bench_driver_args_compact.json
# what one GPU delivers at the per-GPU batch of BASELINE configs[3] as stated on 1 / 2 / 4 / 8 GPUs, the call
# level of the headline batch, the launch gaps of a small batch, the one-step error of the fp32 paths
for prec in fp32-split fp64; do for nb in 64 32 16 8; do python tools/ab_quick.py --batch $nb --precision $prec --iters 200; done; done > $out/${r}_strong_scaling_per_gpu.txt 2>&1
python tools/bench_call.py batch > $out/${r}_call_level.jsonl 2>&1
python tools/call_breakdown.py fp32-split 40 >> $out/${r}_call_level.jsonl 2>&1
python tools/one_step_error.py c5 1 2 3 > $out/${r}_one_step_error.txt 2>&1
python tools/one_step_error.py hl 0 1 2 >> $out/${r}_one_step_error.txt 2>&1
cp $out/trajectory_parity.json $out/${r}_trajectory_parity.json 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && for nb in 1 8; do timeout 300 rocprofv3 --kernel-trace -d /tmp/gap$nb -o trace -- python $GRAFT_REPO_ROOT/tools/profile_target.py --batch $nb --precision fp32-split --iters 300 --streams 1 > /dev/null 2>&1; python $GRAFT_REPO_ROOT/tools/launch_gaps.py /tmp/gap$nb/trace_results.db $GRAFT_REPO_ROOT/$out/${r}_launch_gaps_b$nb.txt > /dev/null; done)
python tools/bench_summary.py $out/${r}_bench_default.json $out/${r}_bench_driver_args.json | cut -c1-200
du -sh $out | tail -1
