for k in 2 3 4 5 6; do
python bench.py --streams $k --no-f64 --no-single --cpu-iters 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('streams', d['config']['streams_per_gpu'], d['value'], d['ms_per_step'], d['timed_region']['ms_per_step_min'], d['timed_region']['ms_per_step_max'])"
done
