#!/bin/bash
# A/B of library builds in ONE call (boxes of the pool differ by several percent): usage  LIBS="a.so b.so" MODE=split REPS=2 tools/ab_bench.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
for rep in $(seq 1 ${REPS:-2}); do
  for lib in $LIBS; do
    if [ "$lib" = default ]; then unset VBX_AMD_LIB; else export VBX_AMD_LIB=$PWD/$lib; fi
    timeout 300 python bench.py --gemm ${MODE:-split} --no-split --no-configs --no-f64 --cpu-iters 0 ${BENCH_ARGS} > gpurun_out/r04_ab.json 2> gpurun_out/r04_ab.err
    python - <<PY
import json
d=json.loads(open('gpurun_out/r04_ab.json').read().strip().splitlines()[-1])
k=d['kernels_avg_us']
print('$lib', round(d['value']), 'rec-it/s', round(d['ms_per_step'],4), 'ms/step; one stream:', round(d.get('one_stream_ms_per_step', 0), 4), {n: round(k[n],1) for n in ('chunk_loglik','chunk_post','mstep_fin','fb_aux')})
PY
  done
done
