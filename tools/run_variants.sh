python -m pytest tests -q -x -m gpu 2>&1 | tail -5
python bench.py 2>&1 | tail -1 > gpurun_out/r02_bench_split.json
cat gpurun_out/r02_bench_split.json
python tools/bench_call.py 2>&1 | tail -4
