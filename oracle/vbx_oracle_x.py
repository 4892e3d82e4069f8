"""Extended-precision referee for the VB-HMM E/M loop  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` (and the committed script tests/golden/make_golden_referee.py) may import this module.

Why it exists.  The reference (``/root/reference/VBx/VBx.py``) works in float64 in the LOG domain: at T = 200 000 its
forward / backward rows reach |lfw| ~ 2e7, where one float64 ulp is 3.7e-9, and the EM map amplifies what the
recursion rounds away.  On two points of BASELINE config 5's sweep -- (Fa, Fb) = (.3, 64) and (.4, 64) after two
iterations -- the fp64 kernels of this repository (scaled LINEAR domain, no large magnitudes) are 1.3e-4 / 1.5e-4 from
the reference's output: above north_star's 1e-4.  Which side is off cannot be told from two float64 programs; it takes
a third computation whose rounding is negligible against both.  This file is that computation:

  * ``VBx_x(..., form='log')``     the reference's algorithm, line for line (VBx.py:87-104 and forward_backward,
                                   VBx.py:146-175: dense S x S log-sum-exp per frame, the 1e-8 epsilons, M-step before
                                   E-step), with every array in ``numpy.longdouble`` -- x87 extended precision on
                                   x86-64 (64-bit significand, eps 1.08e-19: 2048 x finer than float64).  Inputs are the
                                   same float64 values, widened exactly.
  * ``VBx_x(..., form='linear')``  the same model evaluated through the diagonal-plus-rank-one structure of the
                                   transition matrix (as oracle/vbx_oracle.py::fb_linear) in
                                   longdouble: a formulation with a different rounding behaviour altogether.

Two extended-precision evaluations by different routes that agree with each other far below 1e-4 pin the exact result
of the algorithm on these inputs ("truth"); tests/golden/config_c5referee.npz holds it for all nine sweep points
(written by tests/golden/make_golden_referee.py, which also tabulates |reference - truth| from the committed reference
fixture).  The GPU tests then hold every path to north_star's 1e-4 against the truth.

With ``dtype=numpy.float64`` and ``form='log'`` this is oracle/vbx_oracle.py's arithmetic again (tests/test_oracle_golden.py
checks that on small cases), so the referee is pinned to the reference the same way the oracle is.
"""
from __future__ import annotations

import numpy as np

EPS_TR = 1e-8          # VBx.py:158


def _lse(a, axis):
    """max-shifted log-sum-exp in the dtype of ``a`` (scipy.special.logsumexp's definition; VBx.py:24)."""
    m = a.max(axis=axis, keepdims=True)
    s = np.exp(a - m).sum(axis=axis, keepdims=True)
    return np.squeeze(np.log(s) + m, axis=axis)


def forward_backward_x(lls, tr, ip):
    """VBx.py:146-175 in the dtype of ``lls``: (post, tll, lfw, lbw)."""
    dt = lls.dtype
    T = lls.shape[0]
    eps = dt.type(EPS_TR)
    ltr = np.log(tr + eps)                                     # VBx.py:158
    lfw = np.empty_like(lls)
    lbw = np.empty_like(lls)
    lfw[:] = -np.inf
    lbw[:] = -np.inf
    lfw[0] = lls[0] + np.log(ip + eps)                         # VBx.py:163
    lbw[-1] = 0
    ltr_t = np.ascontiguousarray(ltr.T)
    for t in range(1, T):                                      # VBx.py:167-168
        lfw[t] = lls[t] + _lse(lfw[t - 1] + ltr_t, axis=1)
    for t in range(T - 2, -1, -1):                             # VBx.py:170-171
        lbw[t] = _lse(ltr + lls[t + 1] + lbw[t + 1], axis=1)
    tll = _lse(lfw[-1], axis=0)                                # VBx.py:173
    return np.exp(lfw + lbw - tll), tll, lfw, lbw


def _fb_linear_x(lls, pi, lp):
    """The same posteriors, total log-likelihood and the statistic of VBx.py:101-103 through tr + 1e-8 = lp I + 1 c^T
    (c = (1 - lp) pi + 1e-8): O(S) per frame, vectors kept at scale one.  -> (gamma, tll, entered)"""
    dt = lls.dtype
    T, S = lls.shape
    eps = dt.type(EPS_TR)
    c = (1 - lp) * pi + eps
    m = lls.max(axis=1)
    B = np.exp(lls - m[:, None])
    ahat = np.empty_like(lls)
    lsc = np.empty(T, dtype=dt)
    a = B[0] * (pi + eps)
    s = a.sum()
    ahat[0] = a / s
    lsc[0] = np.log(s)
    for t in range(1, T):
        prev = ahat[t - 1]
        a = B[t] * (lp * prev + c * prev.sum())
        s = a.sum()
        ahat[t] = a / s
        lsc[t] = np.log(s)
    tll = lsc.sum() + m.sum()
    bhat = np.empty_like(lls)
    bhat[-1] = 1
    for t in range(T - 2, -1, -1):
        e = B[t + 1] * bhat[t + 1]
        nb = lp * e + np.dot(c, e)
        bhat[t] = nb / nb.max()
    g = ahat * bhat
    g /= g.sum(axis=1, keepdims=True)
    prev = ahat[:-1]
    entered = (g[1:] / (lp * prev + c * prev.sum(axis=1, keepdims=True))).sum(axis=0)     # App. A.4
    return g, tll, entered


def VBx_x(X, Phi, loopProb=0.9, Fa=1.0, Fb=1.0, pi=10, gamma=None, maxIters=10, epsilon=1e-4,
          dtype=np.longdouble, form='log', return_model=False):
    """VBx.py:27-126 in ``dtype`` (no ``ref`` / plotting / model arguments: the referee evaluates fixtures).
    ``gamma`` is required: the global-RNG initialisation is the caller's business."""
    dt = np.dtype(dtype)
    X = np.asarray(X, dtype=dt)
    Phi = np.asarray(Phi, dtype=dt)
    D = X.shape[1]
    if type(pi) is int:                                        # VBx.py:76-77
        pi = np.ones(pi, dtype=dt) / pi
    pi = np.asarray(pi, dtype=dt)
    gamma = np.asarray(gamma, dtype=dt)
    assert gamma.shape[1] == len(pi) and gamma.shape[0] == X.shape[0]
    lp, fa, fb = dt.type(loopProb), dt.type(Fa), dt.type(Fb)
    half = dt.type(0.5)
    G = -half * (np.sum(X ** 2, axis=1, keepdims=True) + D * np.log(2 * dt.type(np.pi)))    # VBx.py:87
    rho = X * np.sqrt(Phi)                                                                   # VBx.py:88-89
    Li = []
    alpha = invL = None
    for ii in range(maxIters):
        invL = 1 / (1 + fa / fb * gamma.sum(axis=0, keepdims=True).T * Phi)                  # VBx.py:95
        alpha = fa / fb * invL * gamma.T.dot(rho)                                            # VBx.py:96
        log_p = fa * (rho.dot(alpha.T) - half * (invL + alpha ** 2).dot(Phi) + G)            # VBx.py:97
        if form == 'log':
            tr = np.eye(len(pi), dtype=dt) * lp + (1 - lp) * pi                              # VBx.py:98
            gamma, log_pX, lfw, lbw = forward_backward_x(log_p, tr, pi)                      # VBx.py:99
            ent = np.exp(_lse(lfw[:-1], axis=1)[:, None] + log_p[1:] + lbw[1:] - log_pX).sum(axis=0)   # VBx.py:101-103
            new_pi = gamma[0] + (1 - lp) * pi * ent
        else:
            gamma, log_pX, ent = _fb_linear_x(log_p, pi, lp)
            new_pi = gamma[0] + (1 - lp) * pi * ent
        elbo = log_pX + fb * half * np.sum(np.log(invL) - invL - alpha ** 2 + 1)             # VBx.py:100
        pi = new_pi / new_pi.sum()                                                           # VBx.py:104
        Li.append([elbo])
        if ii > 0 and elbo - Li[-2][0] < epsilon:                                            # VBx.py:122-125
            break
    out = (gamma, pi, Li)
    return out + (alpha, invL) if return_model else out
