#!/usr/bin/env python
"""Golden fixtures for BASELINE.json's configs at FULL size, generated FROM THE REFERENCE ITSELF.

Runs only in the authoring container (needs /root/reference); the GPU box reads the committed
tests/golden/config_*.npz.  Every fixture holds the arguments of the synthetic generator (vbx_amd.synth) plus
checksums of the regenerated inputs, and the outputs of the unmodified /root/reference/VBx/VBx.py::VBx on them:
pi, the ELBO history, alpha / invL in full (S x D) and gamma on 2000 fixed sampled rows (all rows would be
12-80 MB per case).

  config_c2.npz        C2: T=10 000, S=10, Fa .3 Fb 17 loopProb .99, gamma=None (global RNG, seed 1), 10 iterations
  config_headline.npz  T=10 000, S=30 (the metric's shape): after 2 iterations and after the reference's own stop
                       (maxIters=40, epsilon=1e-4)
  config_c3.npz        C3: T=50 000, S=30, gamma=None (seed 1): after 2, 3 and 40 iterations
  config_c4.npz        C4: recordings 0, 31 and 63 of the 64-recording batch bench.py runs (T=10 000, S=30): 4 iterations
  config_c5.npz        C5: T=200 000, S=50, loopProb .9, sweep points (Fa, Fb) = (.3, 17) and (.2, 6): 2 iterations
  config_c5sweep.npz   C5: all nine points Fa in {.2,.3,.4} x Fb in {6,17,64} (DIHARD2_run.sh:45-46, AMI_run.sh:47,
                       CALLHOME_run.sh:45-46) after 2 iterations (gamma on 500 sampled rows), one process per point
  config_c5stop.npz    C5: the point (.3, 17) run to the reference's own stop (maxIters=40, epsilon=1e-4)

usage: make_golden_configs.py [c2 headline c3 c4 c5]      (no argument: all, one process per config)
"""
import contextlib
import io
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

from vbx_amd.synth import make_recording  # noqa: E402

N_ROWS = 2000


def ref_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_ref_VBx_direct', f'{REF}/VBx/VBx.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sample_rows(T):
    return np.sort(np.random.default_rng(12345).choice(T, size=min(N_ROWS, T), replace=False))


def soft_init(T, S, seed):
    g = np.random.default_rng(seed).gamma(1.0, size=(T, S))
    return g / g.sum(1, keepdims=True)


def run_ref(ref, out, tag, X, Phi, rows, np_seed=None, **kw):
    """One reference call; its outputs go into `out` under tag/..."""
    if np_seed is not None:
        np.random.seed(np_seed)
    t0 = time.time()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        g, p, Li, al, il = ref.VBx(X, Phi, return_model=True, **kw)
    out[tag + '/gamma_rows'] = g[rows]
    out[tag + '/gamma_colsum'] = g.sum(0)
    out[tag + '/pi'] = p
    out[tag + '/Li'] = np.array([r[0] for r in Li])
    out[tag + '/alpha'] = al
    out[tag + '/invL'] = il
    out[tag + '/warned'] = np.asarray('WARNING' in buf.getvalue())
    print(f'  {tag}: {len(Li)} iterations in {time.time() - t0:.0f} s, ELBO {Li[-1][0]:.6f}', flush=True)


def inputs(out, name, T, S, seed, kappa, g0=None):
    X, Phi, _ = make_recording(T, S, seed=seed, kappa=kappa)
    out[name + '/gen'] = np.asarray([T, S, seed, kappa], dtype=np.float64)
    out[name + '/X_checksum'] = np.asarray([X.sum(), (X ** 2).sum(), Phi.sum()])
    if g0 is not None:
        out[name + '/g0_checksum'] = np.asarray([g0.sum(), (g0 ** 2).sum(), g0[T // 2].max()])
    return X, Phi


def c2(ref):
    out = {}
    T, S = 10000, 10
    X, Phi = inputs(out, 'c2', T, S, 0, 0.05)
    rows = sample_rows(T)
    out['c2/rows'] = rows
    out['c2/hyper'] = np.asarray([0.99, 0.3, 17.0])
    run_ref(ref, out, 'c2/it10', X, Phi, rows, np_seed=1, loopProb=0.99, Fa=0.3, Fb=17.0, pi=S, gamma=None,
            maxIters=10, epsilon=-1e300)
    return out


def headline(ref):
    out = {}
    T, S = 10000, 30
    g0 = soft_init(T, S, 1)
    X, Phi = inputs(out, 'hl', T, S, 0, 0.05, g0)
    rows = sample_rows(T)
    out['hl/rows'] = rows
    out['hl/hyper'] = np.asarray([0.99, 0.3, 17.0])
    kw = dict(loopProb=0.99, Fa=0.3, Fb=17.0, pi=S, gamma=g0)
    run_ref(ref, out, 'hl/it2', X, Phi, rows, maxIters=2, epsilon=-1e300, **kw)
    run_ref(ref, out, 'hl/stop', X, Phi, rows, maxIters=40, epsilon=1e-4, **kw)
    return out


def c3(ref):
    out = {}
    T, S = 50000, 30
    X, Phi = inputs(out, 'c3', T, S, 3, 0.05)
    rows = sample_rows(T)
    out['c3/rows'] = rows
    out['c3/hyper'] = np.asarray([0.99, 0.3, 17.0])
    for n in (2, 3, 40):
        run_ref(ref, out, f'c3/it{n}', X, Phi, rows, np_seed=1, loopProb=0.99, Fa=0.3, Fb=17.0, pi=S, gamma=None,
                maxIters=n, epsilon=-1e300)
    return out


def c4(ref):
    out = {}
    T, S = 10000, 30
    rows = sample_rows(T)
    out['c4/rows'] = rows
    out['c4/hyper'] = np.asarray([0.99, 0.3, 17.0])
    for k in (0, 31, 63):                       # bench.py make_batch: recording k = make_recording(seed=k), init seed 10 000 + k
        g0 = soft_init(T, S, 10_000 + k)
        X, Phi = inputs(out, f'c4/rec{k}', T, S, k, 0.05, g0)
        run_ref(ref, out, f'c4/rec{k}/it4', X, Phi, rows, loopProb=0.99, Fa=0.3, Fb=17.0, pi=S, gamma=g0,
                maxIters=4, epsilon=-1e300)
    return out


def c5(ref):
    out = {}
    T, S = 200000, 50
    g0 = soft_init(T, S, 4)
    X, Phi = inputs(out, 'c5', T, S, 3, 0.05, g0)
    rows = sample_rows(T)
    out['c5/rows'] = rows
    for fa, fb in ((0.3, 17.0), (0.2, 6.0)):
        tag = f'c5/fa{fa}_fb{fb:g}'
        out[tag + '/hyper'] = np.asarray([0.9, fa, fb])
        run_ref(ref, out, tag + '/it2', X, Phi, rows, loopProb=0.9, Fa=fa, Fb=fb, pi=S, gamma=g0, maxIters=2,
                epsilon=-1e300)
    return out


C5_GRID = [(fa, fb) for fa in (0.2, 0.3, 0.4) for fb in (6.0, 17.0, 64.0)]


def c5_inputs(out):
    T, S = 200000, 50
    g0 = soft_init(T, S, 4)
    X, Phi = inputs(out, 'c5', T, S, 3, 0.05, g0)
    rows = sample_rows(T)[::4]                              # 500 of the 2000 rows of config_c5.npz
    out['c5/rows'] = rows
    return X, Phi, g0, rows, S


def c5point(ref, k):
    """One point of the nine-point sweep (its own process: 2 x 40 s of reference time each)."""
    out = {}
    X, Phi, g0, rows, S = c5_inputs(out)
    fa, fb = C5_GRID[k]
    tag = f'c5/fa{fa}_fb{fb:g}'
    out[tag + '/hyper'] = np.asarray([0.9, fa, fb])
    run_ref(ref, out, tag + '/it2', X, Phi, rows, loopProb=0.9, Fa=fa, Fb=fb, pi=S, gamma=g0, maxIters=2,
            epsilon=-1e300)
    return out


def c5sweep(ref):
    import subprocess
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), f'c5point{k}']) for k in range(len(C5_GRID))]
    if max(p.wait() for p in procs):
        raise SystemExit('a sweep point failed')
    out = {}
    for k in range(len(C5_GRID)):
        part = os.path.join(HERE, f'config_c5point{k}.npz')
        with np.load(part) as z:
            out.update({key: z[key] for key in z.files})
        os.remove(part)
    return out


def c5stop(ref):
    out = {}
    X, Phi, g0, rows, S = c5_inputs(out)
    out['c5/fa0.3_fb17/hyper'] = np.asarray([0.9, 0.3, 17.0])
    run_ref(ref, out, 'c5/fa0.3_fb17/stop', X, Phi, rows, loopProb=0.9, Fa=0.3, Fb=17.0, pi=S, gamma=g0, maxIters=40,
            epsilon=1e-4)
    return out


CONFIGS = {'c2': c2, 'headline': headline, 'c3': c3, 'c4': c4, 'c5': c5, 'c5sweep': c5sweep, 'c5stop': c5stop}
for _k in range(len(C5_GRID)):
    CONFIGS[f'c5point{_k}'] = (lambda ref, k=_k: c5point(ref, k))


def main():
    names = sys.argv[1:]
    if not names:
        import subprocess
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), n]) for n in CONFIGS
                 if not n.startswith('c5point')]
        sys.exit(max(p.wait() for p in procs))
    ref = ref_module()
    for n in names:
        print(n, flush=True)
        out = CONFIGS[n](ref)
        np.savez_compressed(os.path.join(HERE, f'config_{n}.npz'), **out)


if __name__ == '__main__':
    main()
