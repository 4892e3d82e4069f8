#!/bin/bash
# quick A/B on the GPU box: split tests, the headline in both GEMM modes, optional phase stamps of chunk_post
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
timeout 300 python -m pytest tests/test_gpu_split.py -x -q -m gpu 2>&1 | tail -2
for mode in ${MODES:-exact split}; do
  VBX_AMD_GEMM=$mode timeout 300 python bench.py --no-configs --no-f64 --cpu-iters 0 > gpurun_out/r04_q_$mode.json 2> gpurun_out/r04_q_$mode.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04_q_$mode.json').read().strip().splitlines()[-1])
k=d['kernels_avg_us']
print('$mode', round(d['value']), 'rec-it/s', round(d['ms_per_step'],4), 'ms/step; one stream:', {n: round(k[n],1) for n in ('chunk_loglik','chunk_post','mstep_fin','fb_aux')}, 'single', round(d['single_recording']['ms_per_iteration']*1e3,1),'us')
PY
done
if [ -n "$PHASES" ]; then
  export VBX_AMD_LIB=$PWD/build/libvbx_clk.so
  for p in $PHASES; do echo == $p; timeout 200 python tools/phase_timeline.py 64 10000 30 $p 2>&1 | grep "median phases\|second recorded wave :"; done
fi
