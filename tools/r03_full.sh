#!/bin/bash
# full -m gpu suite + smoke + the default bench line (what the driver runs at round end), timed
export VBX_AMD_NO_REBUILD=1
out=$GRAFT_REPO_ROOT/gpurun_out
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) 2>&1 | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time python bench.py > $out/r03_bench_default.json 2> $out/r03_bench_default.err ) 2>&1 | grep real
( time python bench.py --steps 20 --warmup 5 > $out/r03_bench_driver_args.json 2> $out/r03_bench_driver_args.err ) 2>&1 | grep real
python - <<'PY'
import json
for f in ('r03_bench_default', 'r03_bench_driver_args'):
    d=json.load(open(f'gpurun_out/{f}.json'))
    print(f, 'value', round(d['value']), 'ms', round(d['ms_per_step'],4), 'roof', d['roofline']['kernel'], round(d['roofline']['frac'],3), round(d['roofline']['avg_launch_us'],1), d['roofline']['traffic'], '| f64', round(d['f64']['value']), round(d['f64']['ms_per_step'],4), round(d['f64']['roofline']['frac'],3), '| single', round(d['single_recording']['ms_per_iteration']*1e3,1), 'us')
    for k,v in d['configs'].items(): print('   ', k, round(v['ms_per_iteration'],4), v['dominant_kernel'], round(v['avg_us'],1), round(v['frac'],3), v['traffic'])
PY
