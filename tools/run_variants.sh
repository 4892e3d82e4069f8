for cfg in "8 6" "16 8" "24 12" "32 12" "32 16"; do
set -- $cfg
echo "queues=$1 threads=$2"
for w in "64 1025" "16 4000" "4 10000" "2 20000"; do
set -- $cfg $w
GPU_MAX_HW_QUEUES=$1 VBX_AMD_DRIVER_THREADS=$2 python tools/bench_driver.py --recordings $3 --xvectors $4 --cpu-recordings 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   ', d['config']['workload'][:32], round(d['seconds'],3), d['stages_s'])"
done
done
