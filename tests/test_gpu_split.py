"""The split GEMM mode of the fp32 path (VBX_OPT_GEMM = split: rho alpha^T and gamma^T rho on v_mfma_f32_16x16x32_f16
with error-compensated f16 operand pairs, vbx_amd/csrc/vbx_split.hpp) -- -m gpu.

Held to the SAME bounds as the exact-f32 path (1e-4 on gamma / pi / Li against the reference's outputs: BASELINE.json),
through the same fixtures; and compared with the exact-f32 kernels directly, where the two may differ by rounding only.
The full-size BASELINE configs run under 'fp32-split' in tests/test_gpu_configs.py.
"""
import numpy as np
import pytest

from golden_util import case_inputs

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-4


@pytest.fixture(scope='module')
def ctx():
    from vbx_amd import _capi
    return _capi.Context(0)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def _run(ctx, recs, iters, precision, shared_from=None, epsilon=-np.inf, D=None):
    """recs: list of (X, Phi, g0, lp, Fa, Fb) -> list of result dicts; also returns which GEMM ran."""
    from vbx_amd import _capi
    D = recs[0][0].shape[1]
    batch = _capi.Batch(ctx, [r[0].shape[0] for r in recs], [r[2].shape[1] for r in recs], D, precision=precision, max_iters=iters)
    try:
        if shared_from is not None and batch.streams != 1:
            batch.set_option(_capi.OPT_STREAMS, 1)
        for k, (X, Phi, g0, lp, fa, fb) in enumerate(recs):
            S = g0.shape[1]
            if shared_from is not None and k != shared_from:
                batch.set_recording_shared(k, shared_from, np.ones(S) / S, g0, lp, fa, fb)
            else:
                batch.set_recording(k, X, Phi, np.ones(S) / S, g0, lp, fa, fb)
        batch.run(iters, epsilon)
        return [batch.result(k) for k in range(len(recs))], batch.gemm
    finally:
        batch.close()


def _soft(T, S, seed):
    g = np.random.default_rng(seed).gamma(1.0, size=(T, S))
    return g / g.sum(1, keepdims=True)


def test_split_within_1e4_on_the_fixed_iteration_golden_cases(synth_cases):
    """tests/golden/synth_cases.npz (outputs of the reference itself): the bounds of the exact-f32 test."""
    import vbx_amd
    checked = 0
    for name, c in synth_cases.items():
        if float(c.get('kw_epsilon', 0)) > -1e299 or 'kw_maxIters' not in c or int(c['kw_maxIters']) == 0:
            continue
        X, Phi, kw = case_inputs(c)
        if 'np_seed' in c:
            np.random.seed(int(c['np_seed']))
        gamma, pi, Li, alpha, invL = vbx_amd.VBx(X, Phi, return_model=True, precision='fp32-split', **kw)
        assert len(Li) == len(c['Li']), name
        assert np.abs(gamma - c['gamma']).max() <= FP32_TOL, (name, np.abs(gamma - c['gamma']).max())
        assert np.abs(pi - c['pi']).max() <= FP32_TOL, name
        assert rel_err([r[0] for r in Li], c['Li']) <= FP32_TOL, name
        assert np.abs(alpha - c['alpha']).max() <= FP32_TOL * max(1.0, np.abs(c['alpha']).max()), name
        assert rel_err(invL, c['invL']) <= FP32_TOL, name
        checked += 1
    assert checked >= 9


def test_split_es2005a_thirteen_iterations(ctx, es2005a):
    """The reference's example recording (T = 1025, S = 31, AHC initialisation): 13 iterations as the reference ran them."""
    g = es2005a
    X = g['fea'].astype(np.float32)
    n = len(g['Li40'])
    res, gemm = _run(ctx, [(X, g['Phi'], g['qinit'], float(g['loopProb']), float(g['Fa']), float(g['Fb']))], n, 'fp32-split')
    assert gemm == 'split'
    r = res[0]
    assert len(r['Li']) == n
    assert rel_err(r['Li'], g['Li40']) < 1e-6
    assert np.abs(r['gamma'] - g['gamma40']).max() <= FP32_TOL
    assert np.abs(r['pi'] - g['pi40']).max() <= FP32_TOL
    assert np.array_equal(np.argmax(r['gamma'], 1), np.argmax(g['gamma40'], 1))       # the labels vbhmm.py:160 takes


@pytest.mark.parametrize('T,S,D,scale', [
    (1000, 30, 128, 1.0),        # the bench shape's widths
    (777, 10, 128, 1.0),         # Sp = 16, a ragged last tile
    (1300, 50, 128, 1.0),        # Sp = 64 (the C5 widths)
    (900, 30, 100, 1.0),         # D not a multiple of 32 (padded dims)
    (900, 12, 40, 1.0),          # Dp = 64: two K-blocks
    (900, 30, 160, 1.0),         # Dp = 160: a second slice of the staged model
    (640, 30, 128, 300.0),       # across-class variances x 300^2: rho 300 x larger, alpha 300 x smaller (scale exponents)
    (640, 30, 128, 1 / 30.0),    # ... and the other way round
])
def test_split_against_exact_f32_and_the_oracle(ctx, T, S, D, scale):
    """Three iterations from a soft start: split vs exact-f32 kernels (rounding only) and both vs the float64 oracle."""
    from oracle import vbx_oracle
    from vbx_amd.synth import make_recording
    X, Phi, _ = make_recording(T, S, D=D, seed=T + S, kappa=0.05)
    Phi = Phi * scale ** 2                                 # rho = X sqrt(Phi) (VBx.py:89) scales, the likelihoods stay sane
    g0 = _soft(T, S, 5)
    rec = (X, Phi, g0, 0.95, 0.3, 17.0)
    (rs,), gemm = _run(ctx, [rec], 3, 'fp32-split')
    (rx,), gemm_x = _run(ctx, [rec], 3, 'fp32')
    assert gemm == 'split' and gemm_x == 'exact'
    ref = vbx_oracle.VBx(X, Phi, loopProb=0.95, Fa=0.3, Fb=17.0, pi=S, gamma=g0, maxIters=3, epsilon=-np.inf, return_model=True)
    for r, tag in ((rs, 'split'), (rx, 'exact')):
        assert np.abs(r['gamma'] - ref[0]).max() <= FP32_TOL, (tag, np.abs(r['gamma'] - ref[0]).max())
        assert np.abs(r['pi'] - ref[1]).max() <= FP32_TOL, tag
        assert rel_err(r['Li'], [x[0] for x in ref[2]]) <= 2e-6, tag
    # the two fp32 paths against each other
    assert np.abs(rs['gamma'] - rx['gamma']).max() <= 5e-5
    assert rel_err(rs['Li'], rx['Li']) <= 1e-6
    am = max(1.0, np.abs(rx['alpha']).max())
    assert np.abs(rs['alpha'] - rx['alpha']).max() <= 2e-5 * am


def test_split_sweep_on_a_shared_rho_equals_private_copies(ctx):
    """Recordings that share a rho read their owner's f16 tiles and scale: bit for bit what private copies give."""
    from vbx_amd.synth import make_recording
    T, S = 1500, 30
    X, Phi, _ = make_recording(T, S, seed=9, kappa=0.05)
    g0 = _soft(T, S, 2)
    pts = [(0.9, 0.3, 17.0), (0.9, 0.2, 6.0), (0.8, 0.4, 64.0)]
    recs = [(X, Phi, g0, lp, fa, fb) for lp, fa, fb in pts]
    shared, gemm = _run(ctx, recs, 3, 'fp32-split', shared_from=0)
    assert gemm == 'split'
    private, _ = _run(ctx, recs, 3, 'fp32-split')
    for a, b in zip(shared, private):
        assert np.array_equal(a['gamma'], b['gamma']) and np.array_equal(a['pi'], b['pi']) and np.array_equal(a['Li'], b['Li'])


def test_split_batch_reuse_after_a_recording_is_replaced(ctx):
    """New x-vectors for one recording of a batch: its f16 copies are rebuilt, the others stay."""
    from vbx_amd import _capi
    from vbx_amd.synth import make_recording
    T, S = 700, 12
    data = [make_recording(T, S, seed=20 + k, kappa=0.05)[:2] for k in range(3)]
    g0 = _soft(T, S, 1)
    batch = _capi.Batch(ctx, [T, T], [S, S], 128, precision='fp32-split', max_iters=2)
    try:
        for k in range(2):
            batch.set_recording(k, data[k][0], data[k][1], np.ones(S) / S, g0, 0.9, 0.3, 17.0)
        batch.run(2, -np.inf)
        first = [batch.result(k) for k in range(2)]
        batch.set_recording(1, data[2][0], data[2][1], np.ones(S) / S, g0, 0.9, 0.3, 17.0)
        batch.set_recording(0, data[0][0], data[0][1], np.ones(S) / S, g0, 0.9, 0.3, 17.0)
        batch.run(2, -np.inf)
        assert batch.gemm == 'split'
        second = [batch.result(k) for k in range(2)]
    finally:
        batch.close()
    (alone,), _ = _run(ctx, [(data[2][0], data[2][1], g0, 0.9, 0.3, 17.0)], 2, 'fp32-split')
    assert np.array_equal(first[0]['gamma'], second[0]['gamma'])
    assert np.array_equal(second[1]['gamma'], alone['gamma'])
    assert not np.array_equal(first[1]['gamma'], second[1]['gamma'])


def test_split_is_declined_where_the_fused_kernels_do_not_run(ctx):
    """More than 64 speakers (operators in HBM) and fp64 batches multiply exactly whatever the option says."""
    from vbx_amd import _capi
    from vbx_amd.synth import make_recording
    T, S = 600, 70
    X, Phi, _ = make_recording(T, S, seed=3, kappa=0.05)
    g0 = _soft(T, S, 3)
    (r,), gemm = _run(ctx, [(X, Phi, g0, 0.9, 0.3, 17.0)], 2, 'fp32-split')
    assert gemm == 'exact'
    (rx,), _ = _run(ctx, [(X, Phi, g0, 0.9, 0.3, 17.0)], 2, 'fp32')
    assert np.array_equal(r['gamma'], rx['gamma'])
    batch = _capi.Batch(ctx, [T], [12], 128, precision='fp64', max_iters=1)
    batch.set_option(_capi.OPT_GEMM, _capi.GEMM_SPLIT)
    batch.set_recording(0, X, Phi, np.ones(12) / 12, _soft(T, 12, 1), 0.9, 0.3, 17.0)
    batch.run(1, -np.inf)
    assert batch.gemm == 'exact'
    batch.close()


@pytest.mark.parametrize('precision,S', [('fp32-split', 30), ('fp32', 30), ('fp32-split', 50), ('fp32', 50), ('fp32', 10), ('fp64', 30)])
def test_a_recording_does_not_depend_on_its_batch_at_scale(ctx, precision, S):
    """More workgroups than the chip holds at once (3 x 469 chunks at four per CU, so that workgroups in every phase share
    a CU): a recording's result must not depend on what else is in its batch, nor on the run.  This is the test that the
    wrong sums of DESIGN section 6 fail (a packed-f32 operand form that misreads src1 beside matrix instructions: ~3 of 1400
    chunks, only from the second round of workgroups on): three points on one shared rho with the same hyper-parameters must
    agree with each other and with a single-recording run, bit for bit, after two iterations -- twice."""
    from vbx_amd.synth import make_recording
    T = 60000                                              # (S = 50: the Sp = 64 kernels of C5; S = 10: Sp = 16; fp64: the default path)
    X, Phi, _ = make_recording(T, S, seed=3, kappa=0.05)
    g0 = _soft(T, S, 4)
    rec = (X, Phi, g0, 0.9, 0.3, 17.0)
    (alone,), _ = _run(ctx, [rec], 2, precision)
    for _rep in range(2):
        shared, _ = _run(ctx, [rec, rec, rec], 2, precision, shared_from=0)
        for k, r in enumerate(shared):
            for key in ('gamma', 'pi', 'Li', 'alpha', 'invL'):
                assert np.array_equal(r[key], alone[key]), (precision, _rep, k, key, float(np.abs(np.asarray(r[key]) - np.asarray(alone[key])).max()))


def test_split_declines_x_vectors_one_scale_cannot_cover(ctx):
    """One power-of-two scale per recording carries 22 bits only for frames whose largest element is within 2^10 of the
    recording's largest (vbx_split.hpp: rho_absmax_kernel, kSplitRangeBits).  A recording with a frame 2^-12 of the rest
    makes the batch multiply EXACTLY -- vbx_batch_gemm_in_effect says so and the result is bit for bit the exact path's --
    and new, ordinary x-vectors for that recording bring the split back."""
    from vbx_amd import _capi
    from vbx_amd.synth import make_recording
    T, S = 2000, 9
    X, Phi, _ = make_recording(T, S, seed=21, kappa=0.05, dtype=np.float32)
    g0 = _soft(T, S, 5)
    bad = X.copy()
    bad[777] *= np.float32(2.0 ** -12)
    zero = X.copy()
    zero[5] = 0                                              # (an all-zero frame does not count: it contributes nothing either way)
    res_s, gemm = _run(ctx, [(zero, Phi, g0, 0.9, 0.3, 17.0)], 3, 'fp32-split')
    assert gemm == 'split'
    res_x, gemm = _run(ctx, [(bad, Phi, g0, 0.9, 0.3, 17.0)], 3, 'fp32')
    assert gemm == 'exact'
    res_d, gemm = _run(ctx, [(X, Phi, g0, 0.9, 0.3, 17.0), (bad, Phi, g0, 0.9, 0.3, 17.0)], 3, 'fp32-split')
    assert gemm == 'exact'                                   # declined for the batch: one recording is out of range
    for key in ('gamma', 'pi', 'Li'):
        assert np.array_equal(res_d[1][key], res_x[0][key]), key
    batch = _capi.Batch(ctx, [T], [S], 128, precision='fp32-split', max_iters=3)
    batch.set_recording(0, bad, Phi, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
    batch.run(1, -np.inf)
    assert batch.gemm == 'exact'
    batch.set_recording(0, X, Phi, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
    batch.run(1, -np.inf)
    assert batch.gemm == 'split'
    batch.close()


def test_gemm_environment_variable_is_validated_and_never_overrides_the_argument(ctx, monkeypatch):
    from vbx_amd import _capi
    monkeypatch.setenv('VBX_AMD_GEMM', 'fast')
    with pytest.raises(ValueError, match='VBX_AMD_GEMM'):
        _capi.Batch(ctx, [300], [4], 128, precision='fp32', max_iters=2)
    monkeypatch.setenv('VBX_AMD_GEMM', 'exact')              # the argument names a mode: the environment does not override it
    b = _capi.Batch(ctx, [300], [4], 128, precision='fp32-split', max_iters=2)
    X = np.random.default_rng(0).standard_normal((300, 128)).astype(np.float32)
    b.set_recording(0, X, np.ones(128), np.ones(4) / 4, _soft(300, 4, 1), 0.9, 0.3, 17.0)
    b.run(1, -np.inf)
    assert b.gemm == 'split'
    b.close()
    monkeypatch.setenv('VBX_AMD_GEMM', 'split')              # a plain 'fp32' takes the environment's choice
    b = _capi.Batch(ctx, [300], [4], 128, precision='fp32', max_iters=2)
    b.set_recording(0, X, np.ones(128), np.ones(4) / 4, _soft(300, 4, 1), 0.9, 0.3, 17.0)
    b.run(1, -np.inf)
    assert b.gemm == 'split'
    b.close()
