#!/usr/bin/env python
"""Fixed cost of one vbx_batch_run beside its iterations: warm runs of K = 1, 2, 5, 10, 20, 50, 100 iterations of the headline
batch, least-squares fit t = a + b K.  (The driver's bench uses K = 20: a / 20 is what the fixed cost adds to a step.)"""
import os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch
from vbx_amd import _capi
prec = sys.argv[1] if len(sys.argv) > 1 else 'fp32-split'
streams = int(sys.argv[2]) if len(sys.argv) > 2 else None
ctx = _capi.Context(0)
b = make_batch(ctx, 64, 10000, 30, 128, prec, 0, 4000, streams=streams)
b.run(50, -np.inf)
Ks = [1, 2, 5, 10, 20, 50, 100]
ts = []
for K in Ks:
    t = []
    for _ in range(7):
        t0 = time.perf_counter(); b.run(K, -np.inf); t.append(time.perf_counter() - t0)
    ts.append(sorted(t)[len(t) // 2])
A = np.vstack([np.ones(len(Ks)), Ks]).T
a, s = np.linalg.lstsq(A, np.array(ts), rcond=None)[0]
print(prec, 'streams', b.streams, {K: round(1e3 * t, 3) for K, t in zip(Ks, ts)}, f'fit: fixed {1e3 * a:.3f} ms + {1e3 * s:.4f} ms per iteration')
b.close()
