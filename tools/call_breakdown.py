import os, sys, time, json
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from vbx_amd import _capi
from vbx_amd.synth import make_recording
ctx = _capi.default_context(0)
recs = []
for b in range(64):
    X, Phi, _ = make_recording(10000, 30, seed=b, kappa=0.05, dtype=np.float32)
    g = np.random.default_rng(10_000 + b).gamma(1.0, size=(10000, 30)).astype(np.float32)
    g /= g.sum(1, keepdims=True)
    recs.append((X, Phi, g))
for rep in range(3):
    t0 = time.perf_counter()
    batch = _capi.Batch(ctx, [10000] * 64, [30] * 64, 128, precision='fp32-split', max_iters=10)
    t1 = time.perf_counter()
    for j, (X, Phi, g) in enumerate(recs):
        batch.set_recording(j, X, Phi, np.ones(30) / 30, g, 0.99, 0.3, 17.0)
    t2 = time.perf_counter()
    batch.run(10, -np.inf)
    t3 = time.perf_counter()
    out = [batch.result(j) for j in range(64)]
    t4 = time.perf_counter()
    out2 = [batch.result(j, want_model=False) for j in range(64)]
    t5 = time.perf_counter()
    out3 = [batch.result(j, want_gamma=False, want_model=False) for j in range(64)]
    t6 = time.perf_counter()
    batch.close()
    t7 = time.perf_counter()
    print(json.dumps({'create_ms': 1e3 * (t1 - t0), 'set_ms': 1e3 * (t2 - t1), 'run_ms': 1e3 * (t3 - t2), 'result_all_ms': 1e3 * (t4 - t3),
                      'result_no_model_ms': 1e3 * (t5 - t4), 'result_pi_Li_only_ms': 1e3 * (t6 - t5), 'close_ms': 1e3 * (t7 - t6)}))
