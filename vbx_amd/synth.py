"""Synthetic x-vector sequences for the VB-HMM hot path.

The generator follows the generative assumptions stated in the reference
docstring (/root/reference/VBx/VBx.py:34-36): zero-mean speakers with diagonal
across-class covariance ``Phi`` and identity within-class covariance, plus a
sticky speaker-label chain.  Only numpy is used so that the very same inputs
can be regenerated on the GPU box (bench.py, tests) and in the authoring
container (tests/golden/make_golden.py).

Terminology follows the reference: ``X`` is T x D (frames x PLDA dims),
``Phi`` is the D-vector of across-class variances, S is the number of HMM
states (speakers).
"""
from __future__ import annotations

import numpy as np

__all__ = ["make_phi", "make_recording", "make_lls"]


def make_phi(D: int = 128, seed: int = 0) -> np.ndarray:
    """Descending across-class variances in [0.5, 6.0] (real plda_psi[:128]
    spans 5.60 -> 0.53)."""
    rng = np.random.default_rng(1_000_003 + seed)
    return np.sort(rng.uniform(0.5, 6.0, D))[::-1].copy()


def make_recording(T: int, S: int, D: int = 128, seed: int = 0, kappa: float = 0.05,
                   dwell: float = 50.0, k_true: int | None = None,
                   dtype=np.float64):
    """One synthetic recording.

    Returns ``(X[T,D], Phi[D], labels[T])``.  ``kappa`` scales the speaker
    separation: 1.0 is "easy" (gamma hardens to 0/1 after two iterations),
    0.05 is "soft" (a few percent of frames stay ambiguous, like the real
    ES2005a fixture) and is what parity tests should use.
    """
    rng = np.random.default_rng(seed)
    Phi = make_phi(D, seed=0)          # one PLDA for every recording, as in the recipe
    if k_true is None:
        k_true = min(S, 8)
    k_true = max(1, int(k_true))
    means = rng.standard_normal((k_true, D)) * np.sqrt(kappa * Phi)
    # sticky label chain: switch with prob 1/dwell, new speaker uniform over the others
    switch = rng.random(T) < (1.0 / dwell)
    switch[0] = True
    jump = rng.integers(1, max(k_true, 2), size=T)
    labels = np.empty(T, dtype=np.int64)
    cur = int(rng.integers(0, k_true))
    for t in range(T):
        if switch[t] and k_true > 1 and t > 0:
            cur = (cur + int(jump[t])) % k_true
        labels[t] = cur
    X = means[labels] + rng.standard_normal((T, D))
    return np.ascontiguousarray(X, dtype=dtype), Phi.astype(np.float64), labels


def make_lls(T: int, S: int, seed: int = 0, scale: float = 3.0):
    """Random per-frame state log-likelihoods + a transition matrix of the form the
    reference builds at VBx.py:98, for step-level forward_backward tests."""
    rng = np.random.default_rng(seed)
    lls = scale * rng.standard_normal((T, S)) - 50.0 * rng.random((T, 1))
    pi = rng.random(S) + 0.05
    pi /= pi.sum()
    return lls, pi
