#!/usr/bin/env python
"""tests/golden/config_c5referee.npz: the nine (Fa, Fb) points of BASELINE config 5 after two iterations, evaluated by the
extended-precision referee oracle/vbx_oracle_x.py (numpy.longdouble, the reference's own log-domain algorithm) and
cross-checked against the same model through the linear-domain formulation, also in longdouble.

Needs neither the reference nor a GPU: the inputs are regenerated from the synthetic generator exactly as
make_golden_configs.py::c5_inputs does (checksums compared with the committed reference fixture config_c5sweep.npz), and
the table of |reference - truth| comes from that fixture.  One process per point (4-6 minutes each on one core).

usage: make_golden_referee.py            all nine points, up to $REFEREE_JOBS (default: cores) at a time, then the table
       make_golden_referee.py point<k>   one point -> config_c5referee_point<k>.npz
       make_golden_referee.py table      profiles/r05_c5_referee_reference.json from the two fixtures
"""
import json
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

from vbx_amd.synth import make_recording  # noqa: E402

C5_GRID = [(fa, fb) for fa in (0.2, 0.3, 0.4) for fb in (6.0, 17.0, 64.0)]
T, S = 200000, 50


def tag_of(k):
    fa, fb = C5_GRID[k]
    return f'c5/fa{fa}_fb{fb:g}'


def inputs():
    g0 = np.random.default_rng(4).gamma(1.0, size=(T, S))      # make_golden_configs.py::soft_init(T, S, 4)
    g0 /= g0.sum(1, keepdims=True)
    X, Phi, _ = make_recording(T, S, seed=3, kappa=0.05)
    with np.load(os.path.join(HERE, 'config_c5sweep.npz')) as z:
        assert np.allclose(z['c5/X_checksum'], [X.sum(), (X ** 2).sum(), Phi.sum()], rtol=1e-13), 'not the inputs of the reference fixture'
        assert np.allclose(z['c5/g0_checksum'], [g0.sum(), (g0 ** 2).sum(), g0[T // 2].max()], rtol=1e-13)
        rows = z['c5/rows']
    return X, Phi, g0, rows


def point(k):
    from oracle import vbx_oracle_x
    X, Phi, g0, rows = inputs()
    fa, fb = C5_GRID[k]
    kw = dict(loopProb=0.9, Fa=fa, Fb=fb, pi=S, gamma=g0, maxIters=2, epsilon=-1e300, return_model=True)
    out = {}
    res = {}
    for form in ('log', 'linear'):
        t0 = time.time()
        res[form] = vbx_oracle_x.VBx_x(X, Phi, dtype=np.longdouble, form=form, **kw)
        print(f'  {tag_of(k)} {form}: {time.time() - t0:.0f} s, ELBO {float(res[form][2][-1][0]):.6f}', flush=True)
    g, p, Li, al, il = res['log']
    gl, pl, Lil, all_, ill = res['linear']
    tag = tag_of(k) + '/it2'
    out[tag + '/gamma_rows'] = g[rows].astype(np.float64)
    out[tag + '/gamma_colsum'] = g.sum(0).astype(np.float64)
    out[tag + '/pi'] = p.astype(np.float64)
    out[tag + '/Li'] = np.array([float(r[0]) for r in Li])
    out[tag + '/alpha'] = al.astype(np.float64)
    out[tag + '/invL'] = il.astype(np.float64)
    # how far apart the two extended-precision routes are: the uncertainty of the "truth" (whole gamma, not the sampled rows)
    out[tag + '/forms_disagree'] = np.asarray([float(np.abs(g - gl).max()), float(np.abs(p - pl).max()),
                                               float(max(abs(a[0] - b[0]) / abs(a[0]) for a, b in zip(Li, Lil))),
                                               float(np.abs(al - all_).max()), float(np.abs(il - ill).max())])
    np.savez_compressed(os.path.join(HERE, f'config_c5referee_point{k}.npz'), **out)


def table():
    """|reference - truth| per point, from the two committed fixtures -> profiles/r05_c5_referee_reference.json"""
    doc = {'what': 'BASELINE config 5 (T=200 000, S=50, loopProb 0.9), nine (Fa, Fb) points after two iterations: the '
                   'reference (VBx/VBx.py, float64, log domain; tests/golden/config_c5sweep.npz) against the extended-precision '
                   'referee (oracle/vbx_oracle_x.py, numpy.longdouble, same algorithm; tests/golden/config_c5referee.npz); '
                   'forms_disagree = the referee\'s log-domain and linear-domain evaluations against each other',
           'points': {}}
    with np.load(os.path.join(HERE, 'config_c5sweep.npz')) as ref, np.load(os.path.join(HERE, 'config_c5referee.npz')) as tru:
        for k in range(len(C5_GRID)):
            tag = tag_of(k) + '/it2'
            d = {'gamma_rows_max_abs': float(np.abs(ref[tag + '/gamma_rows'] - tru[tag + '/gamma_rows']).max()),
                 'pi_max_abs': float(np.abs(ref[tag + '/pi'] - tru[tag + '/pi']).max()),
                 'Li_max_rel': float(np.max(np.abs(ref[tag + '/Li'] - tru[tag + '/Li']) / np.abs(tru[tag + '/Li']))),
                 'alpha_max_abs': float(np.abs(ref[tag + '/alpha'] - tru[tag + '/alpha']).max()),
                 'invL_max_rel': float(np.max(np.abs(ref[tag + '/invL'] - tru[tag + '/invL']) / np.abs(tru[tag + '/invL']))),
                 'gamma_colsum_max_rel': float(np.max(np.abs(ref[tag + '/gamma_colsum'] - tru[tag + '/gamma_colsum']) / np.abs(tru[tag + '/gamma_colsum']))),
                 'referee_forms_disagree': dict(zip(('gamma', 'pi', 'Li_rel', 'alpha', 'invL'), map(float, tru[tag + '/forms_disagree'])))}
            doc['points'][tag_of(k)[3:]] = d
    path = os.path.join(REPO, 'profiles', 'r05_c5_referee_reference.json')
    with open(path, 'w') as fh:
        json.dump(doc, fh, indent=1)
    print(json.dumps(doc['points'], indent=1))
    print('wrote', path)


def main():
    arg = sys.argv[1] if len(sys.argv) > 1 else 'all'
    if arg.startswith('point'):
        return point(int(arg[5:]))
    if arg == 'table':
        return table()
    jobs = int(os.environ.get('REFEREE_JOBS', os.cpu_count() or 1))
    pending, running = list(range(len(C5_GRID))), []
    while pending or running:
        while pending and len(running) < jobs:
            k = pending.pop(0)
            running.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), f'point{k}'],
                                            env=dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1')))
        time.sleep(2)
        for p in list(running):
            if p.poll() is not None:
                if p.returncode:
                    raise SystemExit('a point failed')
                running.remove(p)
    out = {}
    for k in range(len(C5_GRID)):
        part = os.path.join(HERE, f'config_c5referee_point{k}.npz')
        with np.load(part) as z:
            out.update({key: z[key] for key in z.files})
        os.remove(part)
    np.savez_compressed(os.path.join(HERE, 'config_c5referee.npz'), **out)
    table()


if __name__ == '__main__':
    main()
