#!/bin/bash
# what the driver runs at round end, on the final tree: the -m gpu suite, smoke(), the bench line (is it one compact line?)
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
( time timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 ) 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 --full-out gpurun_out/final_bench_full.json > gpurun_out/final_bench_compact.json 2> /dev/null
wc -l -c gpurun_out/final_bench_compact.json
python - <<'PY'
import json
c = json.loads(open('gpurun_out/final_bench_compact.json').read().strip().splitlines()[-1])
print({k: c[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'dtype')})
print(c['roofline']); print(c['cpu_baseline'])
f = json.load(open('gpurun_out/final_bench_full.json'))
print(f['hbm_resident_bytes_per_recording'])
PY
