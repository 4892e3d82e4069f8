"""GPU parity of the AHC score stage (vbhmm.py:135-138): cos_similarity and twoGMMcalib_lin through the
C ABI against the reference's golden outputs and the CPU oracle.  float64 end to end; the tolerances are
the rounding of sums taken in a different order."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ahc_cases.npz')


@pytest.fixture(scope='module')
def ahc_cases():
    npz = np.load(GOLDEN)
    out = {}
    for key in npz.files:
        case, field = key.split('/', 1)
        out.setdefault(case, {})[field] = npz[key]
    return out


def test_cos_similarity_and_calibration_match_the_reference(ahc_cases):
    from vbx_amd.diarization_lib import cos_similarity, twoGMMcalib_lin
    for name, c in ahc_cases.items():
        x = c['x'].copy()
        scr = cos_similarity(x)
        assert np.array_equal(x, c['x']), 'input mutated'
        assert scr.dtype == np.float64 and scr.shape == (len(x), len(x))
        np.testing.assert_allclose(scr.ravel()[c['scr_sample_idx']], c['scr_sample'], rtol=0, atol=5e-15, err_msg=name)
        stats = np.array([scr.sum(), (scr ** 2).sum(), scr.min(), scr.max(), np.trace(scr)])
        np.testing.assert_allclose(stats, c['scr_stats'], rtol=1e-12, err_msg=name)
        assert np.abs(scr - scr.T).max() <= 1e-15
        thr, llr = twoGMMcalib_lin(scr.ravel())
        assert isinstance(thr, np.float64) and llr.shape == (scr.size,)
        np.testing.assert_allclose(thr, c['thr'], rtol=1e-10, err_msg=name)
        np.testing.assert_allclose(llr[c['scr_sample_idx']], c['llr_sample'], rtol=1e-9, atol=1e-9, err_msg=name)
        llr_stats = np.array([llr.sum(), (llr ** 2).sum(), llr.min(), llr.max()])
        np.testing.assert_allclose(llr_stats, c['llr_stats'], rtol=1e-9, err_msg=name)
        thr5, _ = twoGMMcalib_lin(scr.ravel().copy(), niters=5)
        np.testing.assert_allclose(thr5, c['thr5'], rtol=1e-10, err_msg=name)


def test_score_matrix_is_an_ordinary_array_and_in_place_edits_reach_the_calibration():
    """The reference returns a plain writable matrix (diarization_lib.py:213) and callers edit it in place; whatever
    they hand to twoGMMcalib_lin afterwards is what gets calibrated."""
    from vbx_amd import diarization_lib as dl
    from oracle import ahc_oracle
    x = np.random.default_rng(0).standard_normal((200, 32))
    scr = dl.cos_similarity(x)
    assert scr.flags.writeable and scr.flags.owndata and scr.flags.c_contiguous
    ref = ahc_oracle.cos_similarity(x)
    np.fill_diagonal(scr, 0.25)                                      # in-place edits of every kind ...
    np.fill_diagonal(ref, 0.25)
    scr *= -1
    ref *= -1
    scr[scr > 0.3] = 0.3
    ref[ref > 0.3] = 0.3
    thr, llr = dl.twoGMMcalib_lin(scr.ravel())                       # ... are what the calibration sees
    thr_o, llr_o = ahc_oracle.twoGMMcalib_lin(ref.ravel())
    np.testing.assert_allclose(thr, thr_o, rtol=1e-9)
    np.testing.assert_allclose(llr, llr_o, rtol=1e-8, atol=1e-8)
    with pytest.raises(ValueError):
        dl.twoGMMcalib_lin(scr)                                      # 2-D: the reference's s[:, np.newaxis] would fail too


@pytest.mark.parametrize('T,D', [(1, 8), (63, 5), (65, 128), (1000, 257)])
def test_shapes_and_edges_against_the_oracle(T, D):
    from vbx_amd.diarization_lib import cos_similarity, twoGMMcalib_lin
    from oracle import ahc_oracle
    x = np.random.default_rng(T + D).standard_normal((T, D)) * 3
    scr = cos_similarity(x)
    np.testing.assert_allclose(scr, ahc_oracle.cos_similarity(x), rtol=0, atol=5e-15)
    if T >= 63:
        thr, llr = twoGMMcalib_lin(scr.ravel(), niters=7)
        thr_o, llr_o = ahc_oracle.twoGMMcalib_lin(ahc_oracle.cos_similarity(x).ravel(), niters=7)
        np.testing.assert_allclose(thr, thr_o, rtol=1e-9)
        np.testing.assert_allclose(llr, llr_o, rtol=1e-8, atol=1e-8)


def test_headline_size_score_stage_properties():
    """T = 10 000 (the headline recording length): 1e8 scores, 800 MB resident.  Oracle-free properties:
    unit diagonal, symmetry, range, and the calibration's fixed point (one more pass does not move it)."""
    from vbx_amd.diarization_lib import cos_similarity, twoGMMcalib_lin
    rng = np.random.default_rng(1)
    centres = rng.standard_normal((6, 128))
    x = centres[rng.integers(0, 6, 10000)] + 0.8 * rng.standard_normal((10000, 128))
    scr = cos_similarity(x)
    np.testing.assert_allclose(np.diag(scr), 1.0, atol=1e-14)
    assert scr.min() >= -1.0001 and scr.max() <= 1.0001                   # diarization_lib.py:211-212
    i, j = rng.integers(0, 10000, 500), rng.integers(0, 10000, 500)
    np.testing.assert_allclose(scr[i, j], scr[j, i], rtol=0, atol=1e-15)
    xn = x / np.linalg.norm(x, axis=1, keepdims=True)
    np.testing.assert_allclose(scr[i, j], np.einsum('kd,kd->k', xn[i], xn[j]), rtol=0, atol=5e-15)
    thr40, _ = twoGMMcalib_lin(scr.ravel(), niters=40)
    thr41, _ = twoGMMcalib_lin(scr.ravel(), niters=41)
    assert abs(thr40 - thr41) < 1e-6 and -1 < thr40 < 1


@pytest.mark.parametrize('T', [1, 2, 3, 64, 65, 257, 1025])
def test_condensed_negated_matrix_is_squareform_of_the_device_matrix(T):
    """vbx_scores_get_condensed(scale = -1) == scipy.spatial.distance.squareform(-scr_mx, checks=False) of the very
    matrix the device holds (vbhmm.py:139): bit for bit, sign of zero included; a vector that is not T x T is refused."""
    from scipy.spatial.distance import squareform
    from vbx_amd import _capi
    ctx = _capi.default_context(None)
    x = np.random.default_rng(T).standard_normal((T, 24))
    x[0] = 0.0                                                   # a zero row: similarities exactly 0 -> -0.0 after negation
    scores = _capi.Scores.cos_similarity(ctx, x)
    try:
        full = scores.get().reshape(T, T)
        cond = scores.get_condensed(T, -1.0)
        want = squareform(-full, checks=False) if T > 1 else np.empty(0)
        assert cond.shape == want.shape and np.array_equal(cond, want)
        assert np.array_equal(np.signbit(cond), np.signbit(want))
        assert np.array_equal(scores.get_condensed(T, 1.0), -want if T > 1 else want)
        if T > 2:
            with pytest.raises(_capi.VbxError):
                scores.get_condensed(T - 1, -1.0)
    finally:
        scores.close()
