for b in 1 2 4 8; do python tools/kbench.py --tag "b$b" --batch $b --iters 100 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print(d['tag'], d['one_stream_ms_per_iter'], d['kernels_us'])"; done
