#!/bin/bash
# rocprofv3 passes over the bench workload (run on the GPU box through gpurun):
#   1. kernel trace (durations)   2. FETCH_SIZE   3. WRITE_SIZE   4. SQ_* (matrix pipes, issue activity)
#   -- counters in their own passes, as the hardware guide prescribes.  The profiled command of passes 1-4 is
#   tools/profile_target.py (the iteration loop only);  5. kernel trace of bench.py itself (same workload and streams,
#   no CPU baseline / fp64 / single-recording legs), whose chunk-kernel averages must agree with the line it prints.
# usage: tools/profile_bench.sh <tag> [profile_target.py args...]      e.g.  tools/profile_bench.sh r02_s1 --streams 1
tag=$1; shift
# (bench.py takes the batch shape and the precision, not profile_target.py's sweep flag)
BENCH_ARGS=$(echo "$*" | sed -e 's/--sweep [a-z]*//')
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
common="python $GRAFT_REPO_ROOT/tools/profile_target.py $*"
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- $common > $out/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o fetch -- $common > $out/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o write -- $common > $out/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY $SQ_EXTRA -d $out/sq -o sq -- $common > $out/sq.log 2>&1
# (optional second SQ pass, SQ2="counter ...": what the SIMDs issue -- SQ_ACTIVE_INST_VALU / _LDS / _SCA beside the matrix pipes)
[ -n "$SQ2" ] && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES $SQ2 -d $out/sq2 -o sq2 -- $common > $out/sq2.log 2>&1
[ -z "$NO_BENCH" ] && timeout 900 rocprofv3 --kernel-trace --stats -d $out/bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-iters 0 --no-f64 --no-single --no-configs $BENCH_ARGS > $out/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $out/trace/trace_results.db $out/kernel_stats.txt > /dev/null
[ -z "$NO_BENCH" ] && python tools/rocpd_stats.py $out/bench/bench_results.db $out/bench_py_kernel_stats.txt > /dev/null
python tools/pmc_counters.py $out/sq/sq_results.db $out/sq_counters.txt > /dev/null
[ -n "$SQ2" ] && python tools/pmc_counters.py $out/sq2/sq2_results.db $out/sq2_counters.txt $out/sq_issue.json $(grep '^workload ' $out/trace.log | cut -d' ' -f2-) > /dev/null
[ -z "$NO_BENCH" ] && grep "^{" $out/bench.log | tail -1 > $out/bench_py_line.json
# HBM bytes per launch, stamped with the workload profile_target.py printed and the commit the library was built from
python tools/pmc_traffic.py $out/fetch/fetch_results.db $out/write/write_results.db $out/pmc_traffic.json $(grep '^workload ' $out/trace.log | cut -d' ' -f2-) > $out/pmc_traffic.txt 2>&1
cat $out/kernel_stats.txt
