python bench.py 2>&1 | tail -1 > gpurun_out/r02_bench_final.json
python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1 > gpurun_out/r02_bench_final_driver_args.json
bash tools/profile_bench.sh r02h_s1 --streams 1 > gpurun_out/prof_r02h_s1.log 2>&1
python tools/pmc_traffic.py gpurun_out/prof_r02h_s1/fetch/fetch_results.db gpurun_out/prof_r02h_s1/write/write_results.db gpurun_out/prof_r02h_s1/pmc_traffic.json batch=64 T=10000 S=30 D=128 precision=fp32 >> gpurun_out/prof_r02h_s1.log 2>&1
rm -rf gpurun_out/prof_r02h_s1/trace gpurun_out/prof_r02h_s1/fetch gpurun_out/prof_r02h_s1/write gpurun_out/prof_r02h_s1/sq gpurun_out/prof_r02h_s1/bench
python tools/bench_call.py 2>&1 | tail -3
