python bench.py 2>&1 | tail -1 > gpurun_out/r02_bench_split.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r02_bench_split_driver_args.json
