"""CPU checks of the chunked-scan algebra (oracle/chunked_scan.py mirrors vbx_amd/csrc/vbx_scan.hpp)
against the reference's log-domain known answers and the O(S) linear restatement."""
import numpy as np
import pytest

from oracle import chunked_scan as cs
from oracle import vbx_oracle as orc
from vbx_amd.synth import make_lls


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-10), (np.float32, 5e-6)])
def test_chunked_model_matches_reference_known_answers(fb_cases, dtype, tol):
    for name, c in fb_cases.items():
        S = c['lls'].shape[1]
        pad = 16
        while pad < S:
            pad *= 2
        g, tll, ent = cs.forward_backward_chunked(c['lls'], c['pi'], float(c['loopProb']), dtype=dtype, pad_to=pad)
        np.testing.assert_allclose(g, c['post'], rtol=0, atol=tol, err_msg=name)
        np.testing.assert_allclose(tll, c['tll'], rtol=1e-12 if dtype is np.float64 else 1e-6, err_msg=name)


def test_zero_columns_and_padded_states_do_not_poison_the_boundary_chain(fb_cases):
    """Regression: a padded state's all-zero backward column once kept a stale exponent, won the
    exponent maximum of the boundary chain and flushed every real weight to zero (NaN gamma)."""
    c = fb_cases['fb_T257_S31']
    args = (c['lls'], c['pi'], float(c['loopProb']))
    with np.errstate(all='ignore'):
        bad, _, _ = cs.forward_backward_chunked(*args, dtype=np.float32, pad_to=32, zero_column_fix=False)
    assert np.isnan(bad).any()                       # the model reproduces the old failure ...
    good, _, _ = cs.forward_backward_chunked(*args, dtype=np.float32, pad_to=32)
    np.testing.assert_allclose(good, c['post'], rtol=0, atol=5e-6)   # ... and the fix removes it


def test_extreme_dynamic_range_and_short_chunks():
    rng = np.random.default_rng(3)
    T, S = 700, 12
    lab = (np.arange(T) // 97) % 5
    lls = -400.0 * rng.random((T, S)) - 300.0
    lls[np.arange(T), lab] = -5.0 * rng.random(T)
    pi = np.ones(S) / S
    ref, tll_ref, ent_ref = orc.fb_linear(lls, pi, 0.9)
    for dtype, tol in ((np.float64, 1e-12), (np.float32, 5e-6)):
        for chunk in (128, 37, 1):
            g, tll, ent = cs.forward_backward_chunked(lls, pi, 0.9, dtype=dtype, chunk=chunk, pad_to=16)
            assert np.all(np.isfinite(g))
            np.testing.assert_allclose(g, ref, rtol=0, atol=tol)
            np.testing.assert_allclose(tll, tll_ref, rtol=1e-6)
    lls, pi = make_lls(300, 5, seed=2, scale=6.0)
    for lp in (0.0, 1.0, 0.5):
        ref, tll_ref, ent_ref = orc.fb_linear(lls, pi, lp)
        g, tll, ent = cs.forward_backward_chunked(lls, pi, lp, chunk=64, pad_to=16)
        np.testing.assert_allclose(g, ref, rtol=0, atol=1e-12)
        np.testing.assert_allclose(ent, ent_ref, rtol=1e-9, atol=1e-12)


def test_subnormal_likelihoods_do_not_overflow_the_rescaling():
    """Regression: b = exp(-95) is a subnormal float32; a column sum that small once produced the
    scale 2^132 = inf.  The exponent is now clamped and the next frame completes the rescaling."""
    rng = np.random.default_rng(5)
    T, S = 400, 6
    lab = (np.arange(T) // 61) % 3
    lls = np.full((T, S), -95.0) - 3.0 * rng.random((T, S))
    lls[np.arange(T), lab] = -rng.random(T)
    pi = np.ones(S) / S
    ref, tll_ref, _ = orc.fb_linear(lls, pi, 0.9)
    with np.errstate(all='ignore'):
        bad, _, _ = cs.forward_backward_chunked(lls, pi, 0.9, dtype=np.float32, pad_to=16, clamp=False)
    good, tll, _ = cs.forward_backward_chunked(lls, pi, 0.9, dtype=np.float32, pad_to=16)
    assert not np.all(np.isfinite(bad))
    assert np.all(np.isfinite(good))
    np.testing.assert_allclose(good, ref, rtol=0, atol=5e-6)
    np.testing.assert_allclose(tll, tll_ref, rtol=1e-6)
