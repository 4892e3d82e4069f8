#!/bin/bash
# the round's closing measurement: profiles (tools/r03_profiles.sh), then the -m gpu suite, smoke and the bench lines
bash tools/r03_profiles.sh > gpurun_out/r03_profiles.log 2>&1
bash tools/collect_profiles.sh > /dev/null      # (so that bench.py finds the PMC files of THIS tree in profiles/)
bash tools/r03_full.sh
python tools/bench_call.py 2>&1 | tail -4
