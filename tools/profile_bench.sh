#!/bin/bash
# rocprofv3 passes over the bench workload (run on the GPU box through gpurun):
#   1. kernel trace (durations)   2. FETCH_SIZE   3. WRITE_SIZE  -- counters in their own passes, as the hardware guide
#   prescribes.  The profiled command is tools/profile_target.py (the iteration loop only).
# usage: tools/profile_bench.sh <tag> [profile_target.py args...]      e.g.  tools/profile_bench.sh r02_s1 --streams 1
tag=$1; shift
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
common="python $GRAFT_REPO_ROOT/tools/profile_target.py $*"
timeout 600 rocprofv3 --kernel-trace --stats -d $out/trace -o trace -- $common > $out/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/fetch -o fetch -- $common > $out/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/write -o write -- $common > $out/write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $out/trace/trace_results.db $out/kernel_stats.txt > /dev/null
cat $out/kernel_stats.txt
