#!/bin/bash
# full -m gpu suite + smoke + the default bench line (what the driver runs at round end), timed
cd "$GRAFT_REPO_ROOT" || exit 1
export VBX_AMD_NO_REBUILD=1
out=gpurun_out
( time timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6 ) 2>&1 | tail -12
cp $out/config_parity.json $out/r04_config_parity.json 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
( time python bench.py > $out/r04_bench_default.json 2> $out/r04_bench_default.err ) 2>&1 | grep real
( time python bench.py --steps 20 --warmup 5 > $out/r04_bench_driver_args.json 2> $out/r04_bench_driver_args.err ) 2>&1 | grep real
python tools/bench_summary.py $out/r04_bench_default.json $out/r04_bench_driver_args.json
