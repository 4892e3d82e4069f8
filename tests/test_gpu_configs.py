"""BASELINE.json's configs at FULL size against outputs of the reference itself (-m gpu).

tests/golden/config_*.npz were written by tests/golden/make_golden_configs.py from the unmodified
/root/reference/VBx/VBx.py::VBx (pi, ELBO history, alpha, invL in full; gamma on 2000 fixed rows + its column sums).
Tolerances: fp64 path 5e-6 absolute on gamma, fp32 path 1e-4 -- the tolerance BASELINE.json's north_star states -- with no
exemption: where the reference itself is further than that from the exact result of its algorithm (two points of the C5 sweep,
REFERENCE_OFF below) the paths are held to the same bounds against an extended-precision referee instead.
The fp64 floor is the reference's own rounding, not the kernels': its log-domain recursion works at |lfw| ~ 1e2 T, so
after ONE iteration its gamma rows miss summing to one by 1e-7 (T = 10 000) to 1e-6 (T = 50 000) and that perturbation
feeds the next M-step -- the first ELBO agrees to 2e-13, the second to 6e-10 (T = 10 000) / 3e-9 (T = 50 000), with the
f64 kernels and with a float64 NumPy model of the linear-domain recursion alike.  Every measured
deviation is also written to gpurun_out/config_parity.json so that the margins can be read off a run.

What "within 1e-4 on fp32" is applied to, quantity by quantity (check() below), with the margins of round 2
(profiles/r02_config_parity.json; the fp32 path keeps N_s = sum_t gamma, every other reduction over T and all ELBO
scalars in f64 from the per-chunk partial sums on -- fin_kernel -- so none of these is an accumulation error):

  gamma, pi         max abs deviation <= 1e-4: the statement of BASELINE.json.  Largest: 8.9e-5 (headline shape after TWO
                    iterations), 3.4e-5 (C3 after two), <= 1e-5 elsewhere, <= 5e-7 on every converged run.  Two or three
                    iterations from a random start is where the EM map itself amplifies a rounding error most (DESIGN
                    section 9: ~25x per iteration while speakers are still forming); the fixtures keep those points on purpose.
  ELBO              relative <= 1e-6 (measured <= 4.4e-7).
  alpha             max abs relative to max |alpha| <= 1e-4 (measured <= 1.4e-5).
  invL              relative <= 1e-4 (round 4: was 2e-4): invL = 1 / (1 + Fa/Fb N_s Phi) carries the deviation of N_s = sum_t
                    gamma of a speaker with little mass relative to ITS mass; measured 9.2e-5 (C3 after three iterations),
                    4.2e-5 (headline, two).
  gamma_colsum_rel  |sum_t gamma - sum_t gamma_ref| / max(1, sum_t gamma_ref) <= 2e-4 on fp32 (round 4: was 4e-4; the largest
                    values measured: 1.16e-4 on C3 after two iterations, 1.6e-4 / 1.8e-4 on the C5 point (.3, 64)).  NOT a per-element figure: a sum over
                    T = 10 000 ... 200 000 per-frame deviations that share a sign while mass is moving between two speakers,
                    divided by the mass of the smaller one (floored at one frame).  Measured 1.16e-4 on C3 after two
                    iterations (a speaker holding ~1 of 50 000 frames is off by 1e-4 frames), 1.2e-5 on the headline shape,
                    <= 1e-5 elsewhere and <= 3e-7 converged.
"""
import json
import os

import numpy as np
import pytest

from golden_util import load_config, config_inputs, config_diffs

pytestmark = pytest.mark.gpu

TOL = {'fp64': 5e-6, 'fp32': 1e-4, 'fp32-split': 1e-4}
# 'fp32-split': fp32 storage, the two GEMMs on f16 operand pairs (VBX_OPT_GEMM = split, vbx_split.hpp) -- held to the very
# bounds of the exact-f32 path
PRECISIONS = ['fp64', 'fp32', 'fp32-split']
_REPORT = {}


@pytest.fixture(scope='module')
def ctx():
    from vbx_amd import _capi
    return _capi.Context(0)


@pytest.fixture(scope='module', autouse=True)
def _write_report():
    yield
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, 'config_parity.json'), 'w') as f:
            json.dump(_REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


# Two fixture points at which the REFERENCE is the outlier: (Fa, Fb) = (.3, 64) and (.4, 64) of the C5 sweep, two iterations from a
# random start on 200 000 frames.  Settled in round 5 by a referee (oracle/vbx_oracle_x.py: the reference's own log-domain
# algorithm in numpy.longdouble, cross-checked against the linear-domain formulation in longdouble -- the two agree to 3e-8;
# tests/golden/config_c5referee.npz, profiles/r05_c5_referee_reference.json): the reference's gamma is 1.32e-4 / 1.47e-4 from the
# exact result of its own algorithm on these inputs (its log-domain recursion rounds at |lfw| ~ 2e7 and the EM map of these points
# multiplies that by ~1e4), at the other seven points 1.2e-6 ... 2.6e-5.  Every path of this repository is therefore held to
# north_star's 1e-4 against the REFEREE at all nine points (check_against_truth); against the reference these two points are
# reported, not bounded -- no implementation can be closer to the reference there than the reference is to the truth.
REFERENCE_OFF = ('c5/fa0.3_fb64/it2', 'c5/fa0.4_fb64/it2')


TRUTH_TOL = {'fp64': 5e-7, 'fp32': 1e-4, 'fp32-split': 1e-4}


def check_against_truth(name, precision, d, n_iters):
    """A path against the extended-precision referee.  The fp32 paths: north_star's 1e-4 on gamma / pi / alpha / invL, no
    exemption (measured over the nine points, profiles/r05_c5_referee_table.md: gamma <= 4.7e-5 exact, <= 3.9e-5 split).
    The fp64 path: 5e-7 -- ten times tighter than its bound against the reference elsewhere, because against the TRUTH the
    reference's own rounding does not enter (measured: gamma <= 3.2e-8, pi <= 7e-10, ELBO <= 5e-12 relative: as close as the
    referee's two formulations are to each other)."""
    _REPORT[f'{name}/{precision}'] = d
    tol = TRUTH_TOL[precision]
    assert d['n_iters'][0] == d['n_iters'][1] == n_iters, (name, precision, d)
    assert d['gamma'] <= tol and d['pi'] <= tol, (name, precision, d)
    assert d['Li_rel'] <= (1e-10 if precision == 'fp64' else 1e-6), (name, precision, d)
    assert d['alpha'] <= tol and d['invL_rel'] <= tol, (name, precision, d)
    assert d['gamma_colsum_rel'] <= (2e-4 if precision != 'fp64' else 4 * tol), (name, precision, d)


def check(name, precision, d, n_iters=None, T=10000):
    _REPORT[f'{name}/{precision}'] = d
    if any(name.startswith(k) for k in REFERENCE_OFF):        # reported only (see REFERENCE_OFF)
        assert d['n_iters'][0] == d['n_iters'][1] == n_iters, (name, precision, d)
        return
    # (the reference's own rounding grows with T: at T = 200 000 its gamma is 7e-6 ... 1.8e-5 from the fp64 kernels after two
    #  iterations -- 2.6e-5 at (Fa, Fb) = (.4, 17), where fp32 is 2.5e-5 and fp32-split 2.3e-5 from it: the deviation is the
    #  reference's -- and 2.0e-5 at its own stop, where fp64, fp32 and fp32-split agree with EACH OTHER to 1e-7; the ELBO of
    #  -1.1e7 agrees to 3.7e-8 there; module docstring)
    tol = TOL[precision] * (max(1.0, T / 25000) if precision == 'fp64' else 1.0)
    if n_iters is not None:
        assert d['n_iters'][0] == n_iters, (name, precision, d)
    assert d['n_iters'][0] == d['n_iters'][1], (name, precision, d)
    assert d['gamma'] <= tol, (name, precision, d)
    assert d['pi'] <= tol, (name, precision, d)
    assert d['Li_rel'] <= (2e-8 * max(1.0, T / 50000) if precision == 'fp64' else 1e-6), (name, precision, d)
    # gamma_colsum_rel is a sum over T per-frame deviations (module docstring): reported, and bounded at 1.5 x the largest
    # values measured (module docstring) rather than at a multiple of the per-element bound
    assert d['gamma_colsum_rel'] <= (2e-4 if precision != 'fp64' else 4 * tol), (name, precision, d)
    if 'alpha' in d:
        # (invL through N_s = sum_t gamma; fp64: twice the gamma bound, the reference's own rounding of N_s at T = 200 000)
        assert d['alpha'] <= tol and d['invL_rel'] <= (2 * tol if precision == 'fp64' else tol), (name, precision, d)


def run_one(ctx, X, Phi, g0, S, hyper, iters, precision, epsilon=-np.inf):
    from vbx_amd import _capi
    lp, fa, fb = (float(v) for v in hyper)
    batch = _capi.Batch(ctx, [X.shape[0]], [S], X.shape[1], precision=precision, max_iters=iters)
    batch.set_recording(0, X, Phi, np.ones(S) / S, g0, lp, fa, fb)
    batch.run(iters, epsilon)
    assert batch.gemm == ('split' if precision == 'fp32-split' else 'exact')      # (the mode asked for is the one that ran)
    res = batch.result(0)
    batch.close()
    return res


@pytest.mark.parametrize('precision', PRECISIONS)
def test_c2_ten_iterations_from_the_global_rng(precision):
    """configs[1]: T=10 000, S=10, gamma=None (VBx.py:79-83 draws it from the global RNG), ten iterations."""
    import vbx_amd
    cfg = load_config('c2')
    X, Phi, _ = config_inputs(cfg, 'c2')
    lp, fa, fb = (float(v) for v in cfg['c2/hyper'])
    np.random.seed(1)
    gamma, pi, Li, alpha, invL = vbx_amd.VBx(X, Phi, loopProb=lp, Fa=fa, Fb=fb, pi=10, gamma=None, maxIters=10,
                                             epsilon=-1e300, return_model=True, precision=precision)
    check('c2/it10', precision, config_diffs(cfg, 'c2/it10', gamma, pi, [r[0] for r in Li], alpha, invL), 10)


@pytest.mark.parametrize('precision', PRECISIONS)
def test_headline_shape_two_iterations_and_converged(ctx, precision):
    """T=10 000, R=128, S=30 (the metric's shape): after two iterations, and the run the reference stops by itself
    (maxIters=40, epsilon=1e-4: seven iterations).  fp64 applies the reference's own stopping rule; fp32 cannot
    resolve 1e-4 on an ELBO of -6e5 and is compared after the same seven iterations."""
    cfg = load_config('headline')
    X, Phi, g0 = config_inputs(cfg, 'hl', g0_seed=1)
    res = run_one(ctx, X, Phi, g0, 30, cfg['hl/hyper'], 2, precision)
    check('hl/it2', precision, config_diffs(cfg, 'hl/it2', res['gamma'], res['pi'], res['Li'], res['alpha'], res['invL']), 2)
    n_ref = len(cfg['hl/stop/Li'])
    if precision == 'fp64':
        res = run_one(ctx, X, Phi, g0, 30, cfg['hl/hyper'], 40, precision, epsilon=1e-4)
    else:
        res = run_one(ctx, X, Phi, g0, 30, cfg['hl/hyper'], n_ref, precision)
    check('hl/stop', precision, config_diffs(cfg, 'hl/stop', res['gamma'], res['pi'], res['Li'], res['alpha'], res['invL']),
          n_ref)


@pytest.mark.parametrize('precision', PRECISIONS)
def test_c3_long_recording(precision):
    """configs[2]: T=50 000, S=30, gamma=None, after 2, 3 and 40 iterations (391 chunks: the two-level boundary walk
    is active)."""
    import vbx_amd
    cfg = load_config('c3')
    X, Phi, _ = config_inputs(cfg, 'c3')
    lp, fa, fb = (float(v) for v in cfg['c3/hyper'])
    for n in (2, 3, 40):
        np.random.seed(1)
        gamma, pi, Li, alpha, invL = vbx_amd.VBx(X, Phi, loopProb=lp, Fa=fa, Fb=fb, pi=30, gamma=None, maxIters=n,
                                                 epsilon=-1e300, return_model=True, precision=precision)
        check(f'c3/it{n}', precision, config_diffs(cfg, f'c3/it{n}', gamma, pi, [r[0] for r in Li], alpha, invL), n, T=50000)


@pytest.mark.parametrize('precision', PRECISIONS)
def test_c4_batch_of_64_on_the_default_streams(ctx, precision):
    """configs[3]: the 64 recordings of T=10 000, S=30 that bench.py runs, as ONE batch on the library's default
    stream groups; recordings 0, 31 and 63 against the reference after four iterations."""
    from vbx_amd import _capi
    from vbx_amd.synth import make_recording
    cfg = load_config('c4')
    T, S, n_rec = 10000, 30, 64
    lp, fa, fb = (float(v) for v in cfg['c4/hyper'])
    batch = _capi.Batch(ctx, [T] * n_rec, [S] * n_rec, 128, precision=precision, max_iters=4)
    assert batch.streams == 3
    for k in range(n_rec):
        if k in (0, 31, 63):
            X, Phi, g0 = config_inputs(cfg, f'c4/rec{k}', g0_seed=10_000 + k)
        else:
            X, Phi, _ = make_recording(T, S, seed=k, kappa=0.05, dtype=np.float32)
            g0 = np.random.default_rng(10_000 + k).gamma(1.0, size=(T, S)).astype(np.float32)
            g0 /= g0.sum(1, keepdims=True)
        batch.set_recording(k, X, Phi, np.ones(S) / S, g0, lp, fa, fb)
    batch.run(4, -np.inf)
    for k in (0, 31, 63):
        res = batch.result(k)
        check(f'c4/rec{k}/it4', precision,
              config_diffs(cfg, f'c4/rec{k}/it4', res['gamma'], res['pi'], res['Li'], res['alpha'], res['invL']), 4)
    batch.close()


@pytest.mark.parametrize('precision', PRECISIONS)
def test_c5_very_long_recording_sweep_points(ctx, precision):
    """configs[4]: T=200 000, S=50, loopProb 0.9, two points of the Fa/Fb sweep as one batch (1563 chunks each:
    two-level walk with groups of 40), two iterations."""
    from vbx_amd import _capi
    cfg = load_config('c5')
    X, Phi, g0 = config_inputs(cfg, 'c5', g0_seed=4)
    points = [(0.3, 17.0), (0.2, 6.0)]
    batch = _capi.Batch(ctx, [X.shape[0]] * 2, [50] * 2, 128, precision=precision, max_iters=2)
    for k, (fa, fb) in enumerate(points):
        batch.set_recording(k, X, Phi, np.ones(50) / 50, g0, 0.9, fa, fb)
    batch.run(2, -np.inf)
    for k, (fa, fb) in enumerate(points):
        tag = f'c5/fa{fa}_fb{fb:g}/it2'
        res = batch.result(k)
        check(tag, precision, config_diffs(cfg, tag, res['gamma'], res['pi'], res['Li'], res['alpha'], res['invL']), 2, T=200000)
    batch.close()


@pytest.mark.parametrize('precision', PRECISIONS)
def test_c5_sweep_on_one_shared_rho(ctx, precision):
    """configs[4] as the library runs a sweep: ONE rho for all points (vbx_batch_set_recording_shared), tiles dealt to the
    XCDs so that the chunks reading the same rows run side by side; the two points the reference computed."""
    from vbx_amd import _capi
    cfg = load_config('c5')
    X, Phi, g0 = config_inputs(cfg, 'c5', g0_seed=4)
    points = [(0.3, 17.0), (0.2, 6.0), (0.4, 64.0)]
    batch = _capi.Batch(ctx, [X.shape[0]] * 3, [50] * 3, 128, precision=precision, max_iters=2)
    batch.set_recording(0, X, Phi, np.ones(50) / 50, g0, 0.9, *points[0])
    for k in (1, 2):
        batch.set_recording_shared(k, 0, np.ones(50) / 50, g0, 0.9, *points[k])
    batch.run(2, -np.inf)
    for k, (fa, fb) in enumerate(points[:2]):
        tag = f'c5/fa{fa}_fb{fb:g}/it2'
        res = batch.result(k)
        check(tag + '/shared', precision, config_diffs(cfg, tag, res['gamma'], res['pi'], res['Li'], res['alpha'], res['invL']), 2, T=200000)
    batch.close()


@pytest.mark.parametrize('precision', PRECISIONS)
def test_c5_all_nine_sweep_points_through_VBx_sweep(precision):
    """configs[4] in full: the nine (Fa, Fb) points of the recipes' grids (DIHARD2_run.sh:45-46, AMI_run.sh:47,
    CALLHOME_run.sh:45-46) on ONE rho per stream through ``VBx_sweep`` -- the call a user of the sweep makes -- after two
    iterations, point by point, against
      * the extended-precision referee (tests/golden/config_c5referee.npz): every point, every path, north_star's bound;
      * the unmodified reference (tests/golden/config_c5sweep.npz, one reference process per point): the seven points where
        the reference itself is within 3e-5 of the referee, under the bounds of every other config; the two where it is not
        (REFERENCE_OFF) are reported."""
    from vbx_amd.batch import VBx_sweep
    cfg = load_config('c5sweep')
    truth = load_config('c5referee')
    truth['c5/rows'] = cfg['c5/rows']
    X, Phi, g0 = config_inputs(cfg, 'c5', g0_seed=4)
    grid = [(fa, fb) for fa in (0.2, 0.3, 0.4) for fb in (6.0, 17.0, 64.0)]
    points = [dict(Fa=fa, Fb=fb, loopProb=0.9, pi=50, gamma=g0) for fa, fb in grid]
    out = VBx_sweep(X, Phi, points, maxIters=2, epsilon=-1e300, precision=precision, return_model=True)
    assert len(out) == 9
    failures = []
    for (fa, fb), (gamma, pi, Li, alpha, invL) in zip(grid, out):
        tag = f'c5/fa{fa}_fb{fb:g}/it2'
        try:                                                # (every point is measured and reported before the test fails)
            check_against_truth(tag + '/truth', precision, config_diffs(truth, tag, gamma, pi, [r[0] for r in Li], alpha, invL), 2)
            check(tag + '/sweep9', precision, config_diffs(cfg, tag, gamma, pi, [r[0] for r in Li], alpha, invL), 2, T=200000)
        except AssertionError as exc:
            failures.append(str(exc)[:400])
    assert not failures, failures


@pytest.mark.parametrize('precision', PRECISIONS)
def test_c5_one_point_to_the_references_own_stop(ctx, precision):
    """configs[4], the point (Fa, Fb) = (.3, 17) run until the reference stops by itself (maxIters=40, epsilon=1e-4): fp64
    applies the stopping rule on the device and must stop at the same iteration; fp32 cannot resolve 1e-4 on an ELBO of
    -1e7 and is compared after the same number of iterations."""
    cfg = load_config('c5stop')
    X, Phi, g0 = config_inputs(cfg, 'c5', g0_seed=4)
    tag = 'c5/fa0.3_fb17/stop'
    n_ref = len(cfg[tag + '/Li'])
    if precision == 'fp64':
        res = run_one(ctx, X, Phi, g0, 50, cfg['c5/fa0.3_fb17/hyper'], 40, precision, epsilon=1e-4)
    else:
        res = run_one(ctx, X, Phi, g0, 50, cfg['c5/fa0.3_fb17/hyper'], n_ref, precision)
    check(tag, precision, config_diffs(cfg, tag, res['gamma'], res['pi'], res['Li'], res['alpha'], res['invL']), n_ref, T=200000)
