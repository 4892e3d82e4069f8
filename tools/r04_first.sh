#!/bin/bash
# round 4, first GPU call: the split-GEMM tests, the parity suite, then the headline in both GEMM modes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export VBX_AMD_NO_REBUILD=1
timeout 600 python -m pytest tests/test_gpu_split.py -x -q -m gpu > gpurun_out/r04_split_tests.log 2>&1
echo "split tests rc=$?" | tee -a gpurun_out/r04_split_tests.log
tail -30 gpurun_out/r04_split_tests.log
timeout 900 python -m pytest tests -q -m gpu --deselect tests/test_gpu_split.py > gpurun_out/r04_gpu_tests.log 2>&1
echo "gpu tests rc=$?" | tee -a gpurun_out/r04_gpu_tests.log
tail -15 gpurun_out/r04_gpu_tests.log
cp gpurun_out/config_parity.json gpurun_out/r04_config_parity_first.json 2>/dev/null
for mode in exact split; do
  VBX_AMD_GEMM=$mode timeout 300 python bench.py --no-configs --no-f64 --cpu-iters 0 > gpurun_out/r04_bench_$mode.json 2> gpurun_out/r04_bench_$mode.err
  echo "bench $mode rc=$?"
  python - <<PY
import json
d=json.loads(open('gpurun_out/r04_bench_$mode.json').read().strip().splitlines()[-1])
print('$mode', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_us'], d.get('kernels_one_stream_us'))
PY
done
