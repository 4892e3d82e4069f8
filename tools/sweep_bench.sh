export VBX_AMD_NO_REBUILD=1
for cfg in "1 10000 30" "8 10000 30" "16 10000 30" "32 10000 30" "64 10000 30" "128 10000 30" "8 10000 10" "1 50000 30" "1 200000 50" "9 200000 50"; do
  set -- $cfg
  echo "== batch=$1 T=$2 S=$3"
  timeout 300 python bench.py --steps 20 --warmup 3 --cpu-iters 0 --no-single --batch $1 --T $2 --S $3 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print({k:d[k] for k in ('value','value_without_any_kernel_events','ms_per_step','device_ms_per_step','kernels_avg_us')}, d['roofline']['kernel'], round(d['roofline']['frac'],3), round(d['roofline_whole_iteration']['frac_of_hbm_peak'],3))"
done
