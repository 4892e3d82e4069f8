#!/usr/bin/env python
"""End-to-end time of ONE reference-style call ``VBx(X, Phi, ...)`` (host arrays in, host arrays out: includes
the PCIe copies, device allocation and the final D2H) next to the device time of its iteration loop."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import vbx_amd
    from vbx_amd.synth import make_recording
    for T, S, iters in ((1025, 31, 13), (10000, 30, 10), (50000, 30, 10)):
        X, Phi, _ = make_recording(T, S, seed=1, kappa=0.05)
        g0 = np.random.default_rng(2).gamma(1.0, size=(T, S))
        g0 /= g0.sum(1, keepdims=True)
        kw = dict(loopProb=0.99, Fa=0.3, Fb=17.0, pi=S, gamma=g0, maxIters=iters, epsilon=-1e300)
        out = {'T': T, 'S': S, 'iterations': iters}
        for precision in ('fp64', 'fp32'):
            vbx_amd.VBx(X, Phi, precision=precision, **kw)                      # warm-up (library load, first launch)
            t = []
            for _ in range(5):
                t0 = time.perf_counter()
                vbx_amd.VBx(X, Phi, precision=precision, **kw)
                t.append(time.perf_counter() - t0)
            out[f'{precision}_call_ms'] = 1e3 * min(t)
            out[f'{precision}_call_iterations_per_s'] = iters / min(t)
        print(json.dumps(out))


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


def batch_call():
    """The headline batch as ONE call with host arrays in and out (VBx_batch: allocation, H2D of 64 x (X, gamma0), the
    iterations, the gamma write-out, D2H of 64 x gamma): the PCIe-inclusive rate beside bench.py's HBM-resident one."""
    from vbx_amd.batch import VBx_batch
    from vbx_amd.synth import make_recording
    recs = []
    for b in range(64):
        X, Phi, _ = make_recording(10000, 30, seed=b, kappa=0.05, dtype=np.float32)
        g = np.random.default_rng(10_000 + b).gamma(1.0, size=(10000, 30)).astype(np.float32)
        g /= g.sum(1, keepdims=True)
        recs.append(dict(X=X, Phi=Phi, pi=30, gamma=g))
    for precision in ('fp32-split', 'fp32', 'fp64'):
        for iters in (10, 40):
            kw = dict(maxIters=iters, epsilon=-1e300, loopProb=0.99, Fa=0.3, Fb=17.0, precision=precision)
            VBx_batch(recs, **kw)
            t = []
            for _ in range(3):
                t0 = time.perf_counter()
                VBx_batch(recs, **kw)
                t.append(time.perf_counter() - t0)
            print(json.dumps({'batch': 64, 'T': 10000, 'S': 30, 'precision': precision, 'iterations': iters, 'call_ms': 1e3 * min(t),
                              'recording_iterations_per_s_call_level': 64 * iters / min(t)}))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'batch':
    batch_call()
