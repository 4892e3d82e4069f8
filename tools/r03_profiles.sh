#!/bin/bash
# The round's profiles (one gpurun call): rocprofv3 kernel trace + PMC FETCH / WRITE + SQ passes of
#   the headline batch, fp32 and fp64, on one stream (with the kernel trace of bench.py itself beside them),
#   the C5 sweep on a shared rho and with private copies (fp32), C2 and C3 (one recording each, fp32).
# Summaries land in gpurun_out/prof_r03_*/ ; tools/collect_profiles.sh copies what is committed into profiles/.
export SQ_EXTRA="SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS"
bash tools/profile_bench.sh r03_f32_s1 --streams 1 > /dev/null 2>&1
bash tools/profile_bench.sh r03_f64_s1 --precision fp64 --streams 1 > /dev/null 2>&1
export NO_BENCH=1
bash tools/profile_bench.sh r03_c5_shared --sweep shared --T 200000 --S 50 > /dev/null 2>&1
bash tools/profile_bench.sh r03_c5_private --sweep private --T 200000 --S 50 > /dev/null 2>&1
bash tools/profile_bench.sh r03_c5_shared_f64 --sweep shared --T 200000 --S 50 --precision fp64 > /dev/null 2>&1
bash tools/profile_bench.sh r03_c2 --batch 1 --T 10000 --S 10 > /dev/null 2>&1
bash tools/profile_bench.sh r03_c3 --batch 1 --T 50000 --S 30 > /dev/null 2>&1
for t in f32_s1 f64_s1 c5_shared c5_private c5_shared_f64 c2 c3; do echo "== $t"; cat gpurun_out/prof_r03_$t/pmc_traffic.txt | grep -v rocclr; grep -E "chunk_|fin|scan" gpurun_out/prof_r03_$t/kernel_stats.txt | cut -c1-110; done
