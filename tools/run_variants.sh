python -m pytest tests -q -x -m gpu 2>&1 | tail -3
python tools/kbench.py --tag cop --iters 100 2>&1 | tail -1
python tools/kbench.py --tag cop-b1 --batch 1 --iters 100 2>&1 | tail -1
python tools/kbench.py --tag cop-f64 --precision fp64 --iters 50 2>&1 | tail -1
python bench.py --no-f64 --no-single --cpu-iters 0 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['kernels_avg_us'], d['roofline']['frac'])"
