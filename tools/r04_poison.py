import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vbx_amd import _capi
from vbx_amd.synth import make_recording
ctx = _capi.Context(0)
for T, S in ((200000, 50), (60000, 30), (10000, 30)):
    X, Phi, _ = make_recording(T, S, seed=3, kappa=0.05)
    g0 = np.random.default_rng(4).gamma(1.0, size=(T, S)); g0 /= g0.sum(1, keepdims=True)
    for prec in ('fp32', 'fp32-split', 'fp64'):
        for n_rec, shared in ((1, False), (3, True)):
            b = _capi.Batch(ctx, [T] * n_rec, [S] * n_rec, 128, precision=prec, max_iters=3)
            if b.streams != 1: b.set_option(_capi.OPT_STREAMS, 1)
            for k in range(n_rec):
                if shared and k: b.set_recording_shared(k, 0, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
                else: b.set_recording(k, X, Phi, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
            b.run(3, -np.inf)
            for k in range(n_rec):
                r = b.result(k)
                bad = {key: int(np.isnan(np.asarray(r[key], dtype=np.float64)).sum()) for key in ('gamma', 'pi', 'Li', 'alpha', 'invL')}
                print(T, S, prec, 'n_rec', n_rec, 'rec', k, 'NaNs', bad, 'Li', r['Li'][-1], flush=True)
            b.close()
