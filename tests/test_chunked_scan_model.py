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


def test_backward_operator_is_the_transpose_of_the_forward_operator():
    """(lp*I + 1 c^T) D_t = (D_t (lp*I + c 1^T))^T, hence B_k = F_k^T: scan1 only builds F_k."""
    rng = np.random.default_rng(0)
    S, L = 7, 40
    B = np.exp(3.0 * rng.normal(size=(L, S)))
    B /= B.max(axis=1, keepdims=True)
    c = 0.01 * rng.random(S) + 1e-8
    F, Ef = cs.scan1(B, c, 0.9, np.float64, False, 0)
    Bk, Eb = cs.scan1(B, c, 0.9, np.float64, False, 1)
    Fm = (F * np.exp2(Ef)[:, None]).T            # Fm[j, i]: column i, row j
    Bm = (Bk * np.exp2(Eb)[:, None]).T
    np.testing.assert_allclose(Bm, Fm.T, rtol=1e-12, atol=0)


def test_padded_states_and_zero_columns_are_harmless(fb_cases):
    """Padded speakers (b = 0, c = 0) create all-zero operator columns; they carry the exponent
    -2^24 so that they can never win the exponent maximum of the boundary chain."""
    c = fb_cases['fb_T257_S31']
    args = (c['lls'], c['pi'], float(c['loopProb']))
    good, _, _ = cs.forward_backward_chunked(*args, dtype=np.float32, pad_to=32)
    np.testing.assert_allclose(good, c['post'], rtol=0, atol=5e-6)
    lls = c['lls'].copy()
    lls[0, 3] = -1e4                              # b_0[3] = 0 exactly: column 3 of chunk 0 is all zero
    ref, _, _ = orc.fb_linear(lls, c['pi'], float(c['loopProb']))
    got, _, _ = cs.forward_backward_chunked(lls, c['pi'], float(c['loopProb']), dtype=np.float32, pad_to=32)
    np.testing.assert_allclose(got, ref, rtol=0, atol=5e-6)


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


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-12), (np.float32, 5e-6)])
@pytest.mark.parametrize('group', [2, 3, 8])
def test_two_level_boundary_walk_model(dtype, tol, group):
    """compose() + group-level walk + in-group walks reproduce the flat chain of mat-vecs (exponents and all)."""
    lls, pi = make_lls(1500, 9, seed=4, scale=5.0)
    lls[0, 2] = -1e4                              # a zero operator column in the very first chunk
    for lp in (0.9, 0.0):
        ref, tll_ref, ent_ref = orc.fb_linear(lls, pi, lp)
        g, tll, ent = cs.forward_backward_chunked(lls, pi, lp, chunk=64, dtype=dtype, pad_to=16, super_group=group)
        np.testing.assert_allclose(g, ref, rtol=0, atol=tol)
        np.testing.assert_allclose(tll, tll_ref, rtol=1e-6)
        np.testing.assert_allclose(ent, ent_ref, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize('dtype,tol', [(np.float64, 1e-12), (np.float32, 5e-6)])
def test_half_chunk_operators_model(dtype, tol):
    """The fused kernels of round 2: operators of the two halves of every chunk, the chunk operator composed from
    them for the boundary walk, the vectors at the cut from one product each -- same gamma, log-likelihood and prior
    statistic as the plain recursion, for full chunks, a tail shorter and a tail longer than half a chunk, a zero
    operator column in the first chunk, with and without the two-level walk."""
    for T in (1500, 1500 + 20, 1500 + 70, 65, 64):
        lls, pi = make_lls(T, 9, seed=5, scale=5.0)
        lls[0, 2] = -1e4
        for lp in (0.9, 0.0):
            ref, tll_ref, ent_ref = orc.fb_linear(lls, pi, lp)
            for group in (1, 3):
                g, tll, ent = cs.forward_backward_chunked(lls, pi, lp, chunk=128, dtype=dtype, pad_to=16,
                                                          super_group=group, split_halves=True)
                np.testing.assert_allclose(g, ref, rtol=0, atol=tol)
                np.testing.assert_allclose(tll, tll_ref, rtol=1e-6)
                np.testing.assert_allclose(ent, ent_ref, rtol=1e-4, atol=1e-6)
