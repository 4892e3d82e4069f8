"""CPU oracle for the score stage of the AHC initialisation  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

float64 NumPy restatement of two functions of ``/root/reference/VBx/diarization_lib.py`` that sit
immediately upstream of ``VBx()`` in ``vbhmm.py:135-138``:

    cos_similarity(x)          diarization_lib.py:190-213   T x T matrix of cosine similarities
    twoGMMcalib_lin(s, niters) diarization_lib.py:13-31     two-Gaussian (shared variance) EM over the
                                                            T*T scores -> threshold, calibrated LLRs

Pinned against outputs of the reference functions themselves (tests/golden/ahc_cases.npz, generated
by tests/golden/make_golden_ahc.py, which imports the reference module).  Only ``tests/`` may import
this module.
"""
from __future__ import annotations

import numpy as np
from scipy.special import softmax


def cos_similarity(x):
    """ref: diarization_lib.py:190-213.  Rows are normalised by (norm + 1e-32); the product is
    accumulated over the feature axis in float64 (the reference does it in slices of that axis only
    to bound memory)."""
    x = np.asarray(x, dtype=np.float64)
    assert x.ndim == 2
    xn = x / (np.sqrt(np.sum(np.square(x), axis=1, keepdims=True)) + 1.0e-32)
    return xn.dot(xn.T)


def twoGMMcalib_lin(s, niters=20):
    """ref: diarization_lib.py:13-31, same statement order (the returned LLRs use the parameters
    the last iteration STARTED with, the threshold the ones it ended with)."""
    s = np.asarray(s, dtype=np.float64)
    weights = np.array([0.5, 0.5])
    means = np.mean(s) + np.std(s) * np.array([-1, 1])
    var = np.var(s)
    threshold = np.inf
    lls = None
    for _ in range(niters):
        lls = np.log(weights) - 0.5 * np.log(var) - 0.5 * (s[:, np.newaxis] - means) ** 2 / var
        gammas = softmax(lls, axis=1)
        cnts = np.sum(gammas, axis=0)
        weights = cnts / cnts.sum()
        means = s.dot(gammas) / cnts
        var = ((s ** 2).dot(gammas) / cnts - means ** 2).dot(weights)
        threshold = -0.5 * (np.log(weights ** 2 / var) - means ** 2 / var).dot([1, -1]) / (means / var).dot([1, -1])
    return threshold, lls[:, means.argmax()] - lls[:, means.argmin()]
