#!/usr/bin/env python
"""Golden fixtures ALONG the EM trajectory: the state after EVERY iteration 1 ... n of BASELINE.json's configs at full size.

The reference's contract is "any maxIters" (VBx.py:27-29: defaults maxIters=10, epsilon=1e-4; the loop VBx.py:91, the stop
test VBx.py:122-125): a caller can stop anywhere, so parity has to hold anywhere.  The fixtures of make_golden_configs.py pin
the two ends (iteration 2 and the stop); these pin everything between.

  traj_<cfg>.npz          outputs of the unmodified /root/reference/VBx/VBx.py::VBx after each iteration k:
                          <cfg>/it<k>/{gamma_rows, gamma_colsum, pi, Li, alpha, invL} (gamma on 400 fixed rows)
  traj_<cfg>_referee.npz  the same from the extended-precision referee oracle/vbx_oracle_x.py (numpy.longdouble, the reference's
                          own log-domain algorithm; `forms_disagree`: against its linear-domain formulation at the last iteration)

How the reference is made to show its trajectory without being modified: the loop's only state is (gamma, pi) -- G, rho, and
every quantity of an iteration are recomputed from them (VBx.py:87-104) -- so a chain of calls with maxIters=1, each handed
the gamma and pi the previous returned, performs exactly the operations of one call with maxIters=n, in the same order on
the same numbers.  That is CHECKED here, bit for bit, against the single-call fixtures config_*.npz wherever they hold the
same iteration (headline 2 and 7, C2 10, C3 2 / 3 / 40, C5 (.3, 17) 12).

  hl   T=10 000, S=30, soft init seed 1 (the metric's shape)     iterations 1 ... 10 (the reference's own stop: 7)
  c2   T=10 000, S=10, gamma=None under np.random.seed(1)         1 ... 10
  c3   T=50 000, S=30, gamma=None under np.random.seed(1)         1 ... 12, 16, 20, 30, 40
  c5   T=200 000, S=50, loopProb .9, (Fa, Fb) = (.3, 17)          1 ... 12 (the reference's own stop)

usage: make_golden_trajectory.py [ref|referee] [hl c2 c3 c5]      (no argument: everything, one process per file)
Needs /root/reference for `ref`; `referee` needs neither the reference nor a GPU.
"""
import contextlib
import io
import os
import subprocess
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

from vbx_amd.synth import make_recording  # noqa: E402

N_ROWS = 400

# name -> (T, S, data seed, g0 seed or None (= global RNG under seed 1), (loopProb, Fa, Fb), iterations stored)
CASES = {
    'hl': (10000, 30, 0, 1, (0.99, 0.3, 17.0), list(range(1, 11))),
    'c2': (10000, 10, 0, None, (0.99, 0.3, 17.0), list(range(1, 11))),
    'c3': (50000, 30, 3, None, (0.99, 0.3, 17.0), list(range(1, 13)) + [16, 20, 30, 40]),
    'c5': (200000, 50, 3, 4, (0.9, 0.3, 17.0), list(range(1, 13))),
}
# single-call fixtures of make_golden_configs.py the chain must reproduce bit for bit: case -> [(file, tag, iteration)]
SINGLE_CALL = {
    'hl': [('headline', 'hl/it2', 2), ('headline', 'hl/stop', 7)],
    'c2': [('c2', 'c2/it10', 10)],
    'c3': [('c3', 'c3/it2', 2), ('c3', 'c3/it3', 3), ('c3', 'c3/it40', 40)],
    'c5': [('c5stop', 'c5/fa0.3_fb17/stop', 12)],
}


def traj_rows(T):
    return np.sort(np.random.default_rng(424242).choice(T, size=N_ROWS, replace=False))


def case_inputs(name):
    T, S, seed, g0_seed, hyper, its = CASES[name]
    X, Phi, _ = make_recording(T, S, seed=seed, kappa=0.05)
    if g0_seed is None:
        np.random.seed(1)                                     # VBx.py:79-83 with alphaQInit = 1
        g0 = np.random.gamma(1.0, size=(T, S))
        g0 = g0 / g0.sum(1, keepdims=True)
    else:
        g0 = np.random.default_rng(g0_seed).gamma(1.0, size=(T, S))
        g0 /= g0.sum(1, keepdims=True)
    return X, Phi, g0


def header(out, name, X, Phi, g0):
    T, S, seed, g0_seed, hyper, its = CASES[name]
    out[name + '/gen'] = np.asarray([T, S, seed, 0.05], dtype=np.float64)
    out[name + '/g0_seed'] = np.asarray(-1 if g0_seed is None else g0_seed)
    out[name + '/X_checksum'] = np.asarray([X.sum(), (X ** 2).sum(), Phi.sum()])
    out[name + '/g0_checksum'] = np.asarray([g0.sum(), (g0 ** 2).sum(), g0[T // 2].max()])
    out[name + '/hyper'] = np.asarray(hyper)
    out[name + '/rows'] = traj_rows(T)
    out[name + '/iterations'] = np.asarray(its)


def store(out, tag, rows, g, p, elbo, al, il):
    out[tag + '/gamma_rows'] = np.asarray(g[rows], dtype=np.float64)
    out[tag + '/gamma_colsum'] = np.asarray(g.sum(0), dtype=np.float64)
    out[tag + '/pi'] = np.asarray(p, dtype=np.float64)
    out[tag + '/Li'] = np.asarray(float(elbo))
    out[tag + '/alpha'] = np.asarray(al, dtype=np.float64)
    out[tag + '/invL'] = np.asarray(il, dtype=np.float64)


def ref_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_ref_VBx_direct', f'{REF}/VBx/VBx.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_ref(name):
    ref = ref_module()
    T, S, seed, g0_seed, (lp, fa, fb), its = CASES[name]
    X, Phi, g0 = case_inputs(name)
    out = {}
    header(out, name, X, Phi, g0)
    rows = out[name + '/rows']
    single = {}
    for fname, tag, k in SINGLE_CALL[name]:
        with np.load(os.path.join(HERE, f'config_{fname}.npz')) as z:
            base = tag.split('/')[0]
            single[k] = {'rows': z[base + '/rows'], 'gamma_rows': z[tag + '/gamma_rows'], 'pi': z[tag + '/pi'],
                         'Li': z[tag + '/Li'], 'alpha': z[tag + '/alpha'], 'tag': tag}
    g, p = g0, np.ones(S) / S                                  # (pi=S: VBx.py:76-77)
    elbos = []
    t0 = time.time()
    for k in range(1, max(its) + 1):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            g, p, Li, al, il = ref.VBx(X, Phi, loopProb=lp, Fa=fa, Fb=fb, pi=p, gamma=g, maxIters=1, epsilon=-1e300,
                                       return_model=True)
        elbos.append(Li[0][0])
        if k in its:
            store(out, f'{name}/it{k}', rows, g, p, Li[0][0], al, il)
        if k in single:                                       # the chain IS the single call: bit for bit
            s = single[k]
            assert np.array_equal(g[s['rows']], s['gamma_rows']), (name, k, 'gamma')
            assert np.array_equal(p, s['pi']) and np.array_equal(al, s['alpha']), (name, k, 'pi / alpha')
            assert np.array_equal(np.asarray(elbos), s['Li'][:k]), (name, k, 'ELBO history')
            print(f'  {name}: iteration {k} of the chain == single call {s["tag"]} bit for bit', flush=True)
        print(f'  {name} ref it{k}: ELBO {Li[0][0]:.6f}  ({time.time() - t0:.0f} s)', flush=True)
    out[name + '/Li_all'] = np.asarray(elbos)
    np.savez_compressed(os.path.join(HERE, f'traj_{name}.npz'), **out)


def run_referee(name):
    from oracle import vbx_oracle_x
    T, S, seed, g0_seed, (lp, fa, fb), its = CASES[name]
    X, Phi, g0 = case_inputs(name)
    out = {}
    header(out, name, X, Phi, g0)
    rows = out[name + '/rows']
    ld = np.longdouble
    g, p = g0.astype(ld), np.ones(S, dtype=ld) / S
    elbos = []
    t0 = time.time()
    last = max(its)
    for k in range(1, last + 1):
        g_in, p_in = g, p
        g, p, Li, al, il = vbx_oracle_x.VBx_x(X, Phi, loopProb=lp, Fa=fa, Fb=fb, pi=p_in, gamma=g_in, maxIters=1,
                                              epsilon=-1e300, dtype=ld, form='log', return_model=True)
        elbos.append(float(Li[0][0]))
        if k in its:
            store(out, f'{name}/it{k}', rows, g, p, Li[0][0], al, il)
        if k == last or k == 3:                               # the other extended-precision route, one step from the same state
            gl, pl, Lil, all_, ill = vbx_oracle_x.VBx_x(X, Phi, loopProb=lp, Fa=fa, Fb=fb, pi=p_in, gamma=g_in, maxIters=1,
                                                        epsilon=-1e300, dtype=ld, form='linear', return_model=True)
            out[f'{name}/it{k}/forms_disagree'] = np.asarray([float(np.abs(g - gl).max()), float(np.abs(p - pl).max()),
                                                              float(abs(Li[0][0] - Lil[0][0]) / abs(Li[0][0]))])
            print(f'  {name} referee it{k}: log vs linear form (one step): gamma {float(np.abs(g - gl).max()):.2e}', flush=True)
        print(f'  {name} referee it{k}: ELBO {float(Li[0][0]):.6f}  ({time.time() - t0:.0f} s)', flush=True)
    out[name + '/Li_all'] = np.asarray(elbos)
    np.savez_compressed(os.path.join(HERE, f'traj_{name}_referee.npz'), **out)


def main():
    args = sys.argv[1:]
    kinds = [a for a in args if a in ('ref', 'referee')] or ['ref', 'referee']
    names = [a for a in args if a in CASES] or list(CASES)
    jobs = [(k, n) for k in kinds for n in names]
    if len(jobs) == 1:
        kind, name = jobs[0]
        return (run_ref if kind == 'ref' else run_referee)(name)
    env = dict(os.environ, OMP_NUM_THREADS='1', OPENBLAS_NUM_THREADS='1')
    # (the reference runs under the BLAS threading the single-call fixtures were made with -- the default -- so that the bit
    #  comparison with them is meaningful; the referee does no BLAS)
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), k, n], env=os.environ if k == 'ref' else env)
             for k, n in jobs]
    sys.exit(max(p.wait() for p in procs))


if __name__ == '__main__':
    main()
