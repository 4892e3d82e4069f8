#!/usr/bin/env python
"""One-iteration error of the fp32 paths at a given point of the EM trajectory: run k iterations on the fp64 path (1e-9 of the
exact result), then ONE more from that state on every path; the deviation from the fp64 step is what a path's own arithmetic adds
in that iteration, free of history.  usage: one_step_error.py [c5|hl|c2|c3] [k ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vbx_amd
from vbx_amd.synth import make_recording
name = sys.argv[1] if len(sys.argv) > 1 else 'c5'
ks = [int(v) for v in sys.argv[2:]] or [1, 2]
T, S, seed, g0_seed, (lp, fa, fb) = {'c5': (200000, 50, 3, 4, (0.9, 0.3, 17.0)), 'hl': (10000, 30, 0, 1, (0.99, 0.3, 17.0)),
                                       'c3': (50000, 30, 3, None, (0.99, 0.3, 17.0)), 'c2': (10000, 10, 0, None, (0.99, 0.3, 17.0))}[name]
X, Phi, _ = make_recording(T, S, seed=seed, kappa=0.05)
if g0_seed is None:
    np.random.seed(1); g0 = np.random.gamma(1.0, size=(T, S)); g0 = g0 / g0.sum(1, keepdims=True)
else:
    g0 = np.random.default_rng(g0_seed).gamma(1.0, size=(T, S)); g0 /= g0.sum(1, keepdims=True)
kw = dict(loopProb=lp, Fa=fa, Fb=fb, epsilon=-1e300)
for k in ks:
    g, p, _ = vbx_amd.VBx(X, Phi, pi=S, gamma=g0, maxIters=k, precision='fp64', **kw)
    step = {prec: vbx_amd.VBx(X, Phi, pi=p, gamma=g, maxIters=1, precision=prec, return_model=True, **kw) for prec in ('fp64', 'fp32', 'fp32-split')}
    for prec in ('fp32', 'fp32-split'):
        dg = np.abs(step[prec][0] - step['fp64'][0])
        da = np.abs(step[prec][3] - step['fp64'][3]).max() / np.abs(step['fp64'][3]).max()
        cs = np.abs(step[prec][0].sum(0) - step['fp64'][0].sum(0)) / np.maximum(1.0, step['fp64'][0].sum(0))
        t, s = np.unravel_index(dg.argmax(), dg.shape)
        print(f'{name}: iteration {k + 1} alone, {prec:10s}: gamma {dg.max():.2e} (frame {t}, speaker {s}; mean {dg.mean():.1e}), pi {np.abs(step[prec][1] - step["fp64"][1]).max():.1e}, '
              f'alpha {da:.1e}, colsum_rel {cs.max():.1e} (speaker {cs.argmax()}, mass {step["fp64"][0].sum(0)[cs.argmax()]:.1f})')
