#!/bin/bash
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_drop_in.py -x -q -m gpu 2>&1 | tail -25
( time timeout 900 python bench.py > $out/r03_bench_a.json 2> $out/r03_bench_a.err ) 2>&1 | tail -3
tail -c 600 $out/r03_bench_a.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03_bench_a.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['kernel'], round(d['roofline']['frac'],3), d['roofline']['traffic_source'])
print('f64', d['f64']['value'], d['f64']['ms_per_step'], d['f64']['roofline']['kernel'], round(d['f64']['roofline']['frac'],3))
for k,v in d['configs'].items(): print(k, round(v['ms_per_iteration'],4), v['dominant_kernel'], round(v['avg_us'],1), round(v['frac'],3), round(v['iteration_frac_of_hbm_peak'],3))
print(d['single_recording'], d['cpu_baseline']['value'])
PY
