#!/usr/bin/env python
"""A quick A/B figure: ms per EM iteration of a batch (default: the headline batch), one vbx_batch_run of --iters iterations after
a warm-up run, on the library VBX_AMD_LIB points to.  usage: ab_quick.py [--batch 64 --T 10000 --S 30 --precision fp32-split --streams N --iters 200]"""
import argparse, os, sys, time
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch, make_sweep_batch
ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=64); ap.add_argument('--T', type=int, default=10000); ap.add_argument('--S', type=int, default=30)
ap.add_argument('--precision', default='fp32-split'); ap.add_argument('--streams', type=int, default=None); ap.add_argument('--iters', type=int, default=200)
ap.add_argument('--sweep', action='store_true'); ap.add_argument('--reps', type=int, default=5)
a = ap.parse_args()
from vbx_amd import _capi
ctx = _capi.Context(0)
if a.sweep:
    b = make_sweep_batch(ctx, a.T, a.S, 128, a.precision, a.iters * (a.reps + 1) + 8, True, streams=a.streams)
    n = 9
else:
    b = make_batch(ctx, a.batch, a.T, a.S, 128, a.precision, 0, a.iters * (a.reps + 1) + 8, streams=a.streams)
    n = a.batch
b.run(a.iters, -np.inf)
ts = []
for _ in range(a.reps):
    t0 = time.perf_counter(); b.run(a.iters, -np.inf); ts.append(time.perf_counter() - t0)
ms = 1e3 * min(ts) / a.iters
print(f'{os.path.basename(os.environ.get("VBX_AMD_LIB", "libvbx_hip.so")):24s} batch={n} T={a.T} S={a.S} {a.precision} streams={b.streams}: '
      f'{ms:.4f} ms/iter (median {1e3 * sorted(ts)[len(ts) // 2] / a.iters:.4f}), {n / ms * 1e3:.0f} rec-it/s')
b.close()
