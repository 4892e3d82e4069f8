#!/bin/bash
# gpurun_out/prof_<tag>/ (scratch) -> profiles/<tag>_* (committed): kernel-trace summary, PMC traffic, SQ counters, and
# for the tags that have it the kernel trace of bench.py itself and the line it printed.
for d in gpurun_out/prof_r03_*; do
  t=$(basename $d); t=${t#prof_}
  [ -f $d/kernel_stats.txt ] && cp $d/kernel_stats.txt profiles/${t}_kernel_stats.txt
  [ -f $d/pmc_traffic.json ] && cp $d/pmc_traffic.json profiles/${t}_pmc_traffic.json
  [ -f $d/sq_counters.txt ] && cp $d/sq_counters.txt profiles/${t}_sq_counters.txt
  [ -f $d/sq2_counters.txt ] && cp $d/sq2_counters.txt profiles/${t}_sq2_counters.txt
  [ -f $d/bench_py_kernel_stats.txt ] && cp $d/bench_py_kernel_stats.txt profiles/${t}_bench_py_kernel_stats.txt
  [ -s $d/bench_py_line.json ] && cp $d/bench_py_line.json profiles/${t}_bench_py_line.json
done
ls profiles | grep r03
