#!/bin/bash
# round 4: after the ds_bpermute fix -- determinism probes, the GPU suite, the headline in both GEMM modes
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
for rep in 1 2 3; do timeout 100 python tools/r04_parts.py 60000 30 3 2>&1 | grep SUMMARY; done
timeout 300 python tools/r04_determinism.py 2>&1 | grep "split" | grep -v "0.00e+00 a 0.00e+00 pi 0.00e+00 L 0.00e+00\]  1-vs-2 recs \[g 0.00e+00 a 0.00e+00 pi 0.00e+00 L 0.00e+00\]  1-vs-3 shared \[g 0.00e+00 a 0.00e+00 pi 0.00e+00 L 0.00e+00\]"
echo determinism done
timeout 900 python -m pytest tests -q -m gpu -x > gpurun_out/r04_gpu_tests2.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r04_gpu_tests2.log
cp gpurun_out/config_parity.json gpurun_out/r04_config_parity_2.json 2>/dev/null
for mode in exact split; do
  VBX_AMD_GEMM=$mode timeout 300 python bench.py --no-configs --no-f64 --cpu-iters 0 > gpurun_out/r04_bench2_$mode.json 2> gpurun_out/r04_bench2_$mode.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04_bench2_$mode.json').read().strip().splitlines()[-1])
print('$mode', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'])
PY
done
