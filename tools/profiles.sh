#!/bin/bash
# The round's profiles in one gpurun call:   tools/profiles.sh <round tag, e.g. r04> [workload tags ...]
# rocprofv3 kernel trace + PMC FETCH_SIZE / WRITE_SIZE + two SQ passes (tools/profile_bench.sh) of
#   f32_s1 / split_s1 / f64_s1   the headline batch (64 x T=10 000 x S=30) on one stream: exact f32, f16 operand pairs, fp64
#                                (with the kernel trace of bench.py itself beside each)
#   c2*, c3*                     BASELINE configs[1], [2] (one recording each): fp32, split, fp64
#   c4x8*                        configs[3] as stated: 8 recordings on this GPU
#   c5_shared*, c5_private_32    configs[4]: the nine-point sweep on one rho; with a private copy per point (fp32)
#   s128*                        one recording with S = 128: the wide chunked scan (vbx_scan_wide.hpp)
# Summaries land in gpurun_out/prof_<round>_*/ and are copied into profiles/<round>_* on the box and into
# gpurun_out/profiles_<round>/ (what comes back: copy that directory's files into profiles/; the files bench.py reads carry the
# hash of the kernel sources: a later change of the kernels retires them).
r=$1; shift
want="$*"
export SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES"
export SQ_EXTRA="SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS"
run() {  # tag, NO_BENCH flag, args...
  tag=$1; nb=$2; shift 2
  if [ -n "$want" ] && ! echo " $want " | grep -q " $tag "; then return; fi
  NO_BENCH=$nb bash tools/profile_bench.sh ${r}_$tag "$@" > /dev/null 2>&1
  d=gpurun_out/prof_${r}_$tag
  mkdir -p gpurun_out/profiles_$r
  for f in kernel_stats.txt pmc_traffic.json sq_counters.txt sq2_counters.txt sq_issue.json bench_py_kernel_stats.txt bench_py_line.json; do
    [ -s $d/$f ] && cp $d/$f profiles/${r}_${tag}_$f && cp $d/$f gpurun_out/profiles_$r/${r}_${tag}_$f
  done
  rm -rf $d/trace $d/fetch $d/write $d/sq $d/sq2 $d/bench      # (the rocprofv3 databases: gpurun_out/ travels back only below 64 MiB)
  echo "== $tag"; grep -v rocclr $d/pmc_traffic.txt 2>/dev/null | grep -E "chunk_|fin|scan"; grep -E "chunk_|fin_|scan" $d/kernel_stats.txt | cut -c1-120
}
run f32_s1 "" --streams 1
run split_s1 "" --precision fp32-split --streams 1
run f64_s1 "" --precision fp64 --streams 1
for p in fp32 fp32-split fp64; do
  s=$(echo $p | sed -e 's/fp32-split/split/' -e 's/fp//')
  run c2_$s 1 --batch 1 --T 10000 --S 10 --precision $p
  run c3_$s 1 --batch 1 --T 50000 --S 30 --precision $p
  run c4x8_$s 1 --batch 8 --T 10000 --S 30 --precision $p --streams 1
  run c5_shared_$s 1 --sweep shared --T 200000 --S 50 --precision $p
done
run c5_private_32 1 --sweep private --T 200000 --S 50 --precision fp32
run s128_32 1 --batch 1 --T 10000 --S 128 --precision fp32
run s128_64 1 --batch 1 --T 10000 --S 128 --precision fp64
ls profiles | grep "^${r}_" | wc -l
