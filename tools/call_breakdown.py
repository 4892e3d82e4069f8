#!/usr/bin/env python
"""Where one VBx_batch call of the headline batch (64 recordings of T = 10 000, host arrays in and out) spends its time:
create / enqueue the uploads / wait for them / the iterations / fetch the results / close -- the ABI-7 path
(VBX_OPT_ASYNC_UPLOAD, vbx_batch_get_results into pinned memory) beside the call-by-call path of ABI 6.
usage: call_breakdown.py [precision=fp32-split] [iterations=40]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vbx_amd import _capi
from vbx_amd.synth import make_recording

precision = sys.argv[1] if len(sys.argv) > 1 else 'fp32-split'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = _capi.default_context(0)
recs = []
for b in range(64):
    X, Phi, _ = make_recording(10000, 30, seed=b, kappa=0.05, dtype=np.float32)
    g = np.random.default_rng(10_000 + b).gamma(1.0, size=(10000, 30)).astype(np.float32)
    g /= g.sum(1, keepdims=True)
    recs.append((X, Phi, g))
pi0 = np.ones(30) / 30
for mode in ('abi7', 'abi7', 'abi7', 'abi6', 'abi6'):
    t = [time.perf_counter()]
    batch = _capi.Batch(ctx, [10000] * 64, [30] * 64, 128, precision=precision, max_iters=iters)
    t.append(time.perf_counter())
    if mode == 'abi7':
        batch.set_async_upload(True)
        from concurrent.futures import ThreadPoolExecutor
        by = {}
        for j in range(64):
            by.setdefault(batch.stream_of(j), []).append(j)
        def up(js):
            for j in js:
                X, Phi, g = recs[j]
                batch.set_recording(j, X, Phi, pi0, g, 0.99, 0.3, 17.0)
        with ThreadPoolExecutor(max_workers=len(by)) as pool:
            for f in [pool.submit(up, js) for js in by.values()]:
                f.result()
    else:
        for j, (X, Phi, g) in enumerate(recs):
            batch.set_recording(j, X, Phi, pi0, g, 0.99, 0.3, 17.0)
    t.append(time.perf_counter())
    if mode == 'abi7':
        batch.sync_uploads()
    t.append(time.perf_counter())
    batch.run(iters, -np.inf)
    t.append(time.perf_counter())
    out = batch.results() if mode == 'abi7' else [batch.result(j) for j in range(64)]
    t.append(time.perf_counter())
    batch.close()
    t.append(time.perf_counter())
    names = ('create', 'set', 'sync_uploads', 'run', 'results', 'close')
    print(json.dumps(dict({'mode': mode, 'precision': precision, 'iterations': iters, 'total_ms': round(1e3 * (t[-1] - t[0]), 2)},
                          **{n + '_ms': round(1e3 * (b - a), 2) for n, a, b in zip(names, t[:-1], t[1:])})))
    del out
