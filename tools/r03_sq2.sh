#!/bin/bash
# second SQ pass of the headline batch (fp32, fp64) and the C5 sweep: vector-issue share of the SIMD cycles beside the matrix pipes
export SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"
export NO_BENCH=1
bash tools/profile_bench.sh r03_f32_s1 --streams 1 > /dev/null 2>&1
bash tools/profile_bench.sh r03_f64_s1 --precision fp64 --streams 1 > /dev/null 2>&1
bash tools/profile_bench.sh r03_c5_shared --sweep shared --T 200000 --S 50 > /dev/null 2>&1
for t in f32_s1 f64_s1 c5_shared; do echo "== $t"; tail -12 gpurun_out/prof_r03_$t/sq2_counters.txt | cut -c1-260; done
