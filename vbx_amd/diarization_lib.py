"""Host-side mirror of the score stage of the reference's AHC initialisation on top of libvbx_hip.so.

``vbhmm.py:135-138`` computes, right before it calls ``VBx()``::

    scr_mx = cos_similarity(x)                       # diarization_lib.py:190-213, T x T float64
    thr, _ = twoGMMcalib_lin(scr_mx.ravel())         # diarization_lib.py:13-31, 20 EM passes over T*T scores

Both functions keep the reference's signatures, return types and error behaviour; the work runs in HIP
kernels (f64 MFMA for the similarity matrix, one streaming kernel per EM pass).  ``cos_similarity`` returns an
ordinary, writable ndarray like the reference's (callers edit it in place: ``np.fill_diagonal``, ``scr_mx *= -1``,
masks), and ``twoGMMcalib_lin`` uploads whatever vector it is handed: the host array is the only truth, so an in-place
edit between the two calls cannot go unnoticed.  (Round 2 kept the device copy of the matrix alive behind a read-only
array and spot-checked it; a full check costs more than the 8 T^2 bytes over PCIe it saved.)  Callers that want the
matrix to stay in HBM between the two steps use ``vbx_amd._capi.Scores`` directly, as ``vbx_amd.vbhmm`` does.
There is no CPU fallback.
"""
from __future__ import annotations

import numpy as np

from . import _capi

__all__ = ['cos_similarity', 'twoGMMcalib_lin']


def cos_similarity(x, *, device=None):
    """Cosine similarity matrix of the rows of ``x`` (T x D) -> T x T float64.  diarization_lib.py:190-213."""
    x = np.asarray(x)
    assert x.ndim == 2, f'x has {x.ndim} dimensions, it must be matrix'
    x = np.ascontiguousarray(x, dtype=np.float64)
    # the reference asserts that every row normalises to unit length (diarization_lib.py:201-202)
    norm = np.sqrt(np.sum(np.square(x), axis=1, keepdims=True))
    xn_sq = np.sum(np.square(x / (norm + 1.0e-32)), axis=1)
    assert np.allclose(np.ones_like(xn_sq), xn_sq)
    scores = _capi.Scores.cos_similarity(_capi.default_context(device), x)
    try:
        out = np.empty((x.shape[0], x.shape[0]))
        scores.get(out=out)
    finally:
        scores.close()
    return out


def twoGMMcalib_lin(s, niters=20, *, device=None):
    """Two-Gaussian GMM with shared variance over the scores ``s``: returns the threshold that separates
    the two Gaussians and the linearly calibrated log-odds of every score.  diarization_lib.py:13-31."""
    s = np.asarray(s)
    if s.ndim != 1:
        raise ValueError('twoGMMcalib_lin expects a vector of scores')     # the reference's s[:, np.newaxis] needs 1-D
    scores = _capi.Scores.upload(_capi.default_context(device), s)
    try:
        threshold, llr = scores.two_gmm_calib(niters)
    finally:
        scores.close()
    return np.float64(threshold), llr
