python tools/bench_driver.py --recordings 64 --xvectors 1025 2>&1 | tail -1
python tools/bench_driver.py --recordings 16 --xvectors 4000 --cpu-recordings 0 2>&1 | tail -1
python tools/bench_driver.py --recordings 4 --xvectors 10000 --cpu-recordings 0 2>&1 | tail -1
python tools/bench_driver.py --recordings 2 --xvectors 20000 --cpu-recordings 0 2>&1 | tail -1
