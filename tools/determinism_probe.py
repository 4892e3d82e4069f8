"""Is a recording's result independent of what else is in its batch, and of the run?  (split vs exact GEMM)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vbx_amd import _capi
from vbx_amd.synth import make_recording

ctx = _capi.Context(0)


def run(T, S, prec, n_rec, iters, shared=False, which=0):
    X, Phi, _ = make_recording(T, S, seed=3, kappa=0.05)
    g0 = np.random.default_rng(4).gamma(1.0, size=(T, S)); g0 /= g0.sum(1, keepdims=True)
    pts = [(0.3, 17.0), (0.2, 6.0), (0.4, 64.0), (0.2, 17.0)][:n_rec]
    b = _capi.Batch(ctx, [T] * n_rec, [S] * n_rec, 128, precision=prec, max_iters=iters)
    if b.streams != 1:
        b.set_option(_capi.OPT_STREAMS, 1)
    for k, (fa, fb) in enumerate(pts):
        if shared and k:
            b.set_recording_shared(k, 0, np.ones(S) / S, g0, 0.9, fa, fb)
        else:
            b.set_recording(k, X, Phi, np.ones(S) / S, g0, 0.9, fa, fb)
    b.run(iters, -np.inf)
    r = b.result(which)
    b.close()
    return r


for T, S in ((3000, 50), (20000, 50), (60000, 50), (200000, 50), (60000, 30)):
    for prec in ('fp32', 'fp32-split'):
        for iters in (1, 2):
            a = run(T, S, prec, 1, iters)
            a2 = run(T, S, prec, 1, iters)
            b = run(T, S, prec, 2, iters)
            c = run(T, S, prec, 3, iters, shared=True)
            def d(x, y):
                return f"g {np.abs(x['gamma'] - y['gamma']).max():.2e} a {np.abs(x['alpha'] - y['alpha']).max():.2e} pi {np.abs(x['pi'] - y['pi']).max():.2e} L {abs(x['Li'][-1] - y['Li'][-1]):.2e}"
            print(f'T={T} S={S} {prec:10s} it={iters}: rerun [{d(a, a2)}]  1-vs-2 recs [{d(a, b)}]  1-vs-3 shared [{d(a, c)}]', flush=True)
