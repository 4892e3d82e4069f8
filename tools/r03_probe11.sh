#!/bin/bash
export VBX_AMD_NO_REBUILD=1
python - <<'PY'
import os, sys, time, json
sys.path.insert(0, '.')
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np
from vbx_amd import _capi
from vbx_amd.synth import make_recording
from bench import SWEEP_POINTS
ctx = _capi.Context(0)
T, S = 200000, 50
X, Phi, _ = make_recording(T, S, D=128, seed=0, kappa=0.05, dtype=np.float32)
g = np.random.default_rng(10_000).gamma(1.0, size=(T, S)).astype(np.float32); g /= g.sum(1, keepdims=True)
for streams in (1, 2, 3):
    os.environ['VBX_AMD_STREAMS'] = str(streams)
    n = len(SWEEP_POINTS)
    b = _capi.Batch(ctx, [T] * n, [S] * n, 128, precision='fp32', max_iters=80)
    for k, (fa, fb) in enumerate(SWEEP_POINTS):
        b.set_recording(k, X, Phi, np.ones(S) / S, g, 0.9, fa, fb)
    b.run(6, -np.inf)
    t0 = time.perf_counter(); b.run(20, -np.inf); dt = time.perf_counter() - t0
    print('private copies, streams', b.streams, 'ms per iteration', round(1e3 * dt / 20, 4), flush=True)
    b.close()
PY
