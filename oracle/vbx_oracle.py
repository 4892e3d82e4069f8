"""CPU oracle for the VB-HMM E/M hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and only as the checker.  The shipped path
(``vbx_amd``) never imports it and fails loudly when its HIP library is absent.

This file is a float64 NumPy/SciPy restatement of the algorithm of
``/root/reference/VBx/VBx.py`` (the ``VBx`` function, lines 27-126, its helper
``forward_backward``, lines 146-175, and ``DER``, lines 134-143).  It keeps the
reference's arithmetic order where that order is visible in the results
(log-domain recursions with ``scipy.special.logsumexp``, the 1e-8 epsilons,
M-step before E-step, the convergence test), so that it is also a fair stand-in
for the reference's CPU cost when timed (bench.py ``cpu_baseline``,
``kind="port"``).

Parity pinning: the reference repository has no tests and no golden vectors for
this path.  The oracle is therefore pinned against outputs of
the reference itself, generated in the authoring container by
``tests/golden/make_golden.py`` (which imports /root/reference/VBx/VBx.py) and
committed under ``tests/golden/`` -- see ``tests/test_oracle_golden.py``.

The second half of the file (``fb_linear`` and friends) restates the same
forward-backward in the scaled linear domain using the rank-one-plus-diagonal
structure of the transition matrix.  The HIP kernels are
built on that form; having it here lets the CPU test-suite prove the algebra
against the log-domain restatement without a GPU.
"""
from __future__ import annotations

import numpy as np
from scipy.special import logsumexp

LOG_2PI = float(np.log(2.0 * np.pi))
EPS_TR = 1e-8          # VBx.py:158  (added to tr and to ip before taking logs)


# --------------------------------------------------------------------------------------
# forward-backward, log domain                                   ref: VBx.py:146-175
# --------------------------------------------------------------------------------------
def forward_backward(lls, tr, ip):
    """State posteriors of an HMM given per-frame state log-likelihoods.

    lls[T,S], tr[S,S] (row = from-state), ip[S]  ->  (post[T,S], tll, lfw[T,S], lbw[T,S])
    ref: VBx.py:146-175 (eps VBx.py:158; init :163-165; fwd :167-168; bwd :170-171;
    total and posteriors :173-174).
    """
    lls = np.asarray(lls)
    n_frames = lls.shape[0]
    log_tr = np.log(tr + EPS_TR)
    lfw = np.full_like(lls, -np.inf)
    lbw = np.full_like(lls, -np.inf)
    lfw[0] = lls[0] + np.log(ip + EPS_TR)
    lbw[n_frames - 1] = 0.0
    log_tr_t = log_tr.T
    for t in range(1, n_frames):
        # lfw[t, j] = lls[t, j] + log sum_i exp(lfw[t-1, i] + log_tr[i, j])
        lfw[t] = lls[t] + logsumexp(lfw[t - 1] + log_tr_t, axis=1)
    for t in range(n_frames - 2, -1, -1):
        # lbw[t, i] = log sum_j exp(log_tr[i, j] + lls[t+1, j] + lbw[t+1, j])
        lbw[t] = logsumexp(log_tr + lls[t + 1] + lbw[t + 1], axis=1)
    tll = logsumexp(lfw[n_frames - 1], axis=0)
    post = np.exp(lfw + lbw - tll)
    return post, tll, lfw, lbw


# --------------------------------------------------------------------------------------
# pieces of one VB iteration                                       ref: VBx.py:87-104
# --------------------------------------------------------------------------------------
def frame_constants(X, Phi):
    """G[T,1] and rho[T,D], computed once per call.  ref: VBx.py:87-89."""
    D = X.shape[1]
    G = -0.5 * (np.sum(X ** 2, axis=1, keepdims=True) + D * LOG_2PI)
    rho = X * np.sqrt(Phi)
    return G, rho


def speaker_model(gamma, rho, Phi, Fa, Fb):
    """M-step: (alpha[S,D], invL[S,D]).  ref: VBx.py:95-96 (eqs 16, 17)."""
    occupancy = gamma.sum(axis=0, keepdims=True).T           # N_s as a column
    invL = 1.0 / (1 + Fa / Fb * occupancy * Phi)
    alpha = Fa / Fb * invL * gamma.T.dot(rho)
    return alpha, invL


def frame_loglik(rho, alpha, invL, Phi, G, Fa):
    """log_p[T,S].  ref: VBx.py:97 (eq 23)."""
    return Fa * (rho.dot(alpha.T) - 0.5 * (invL + alpha ** 2).dot(Phi) + G)


def transition_matrix(pi, loopProb):
    """tr[S,S]; column j receives (1-loopProb)*pi_j, diagonal adds loopProb.  ref: VBx.py:98."""
    return np.eye(len(pi)) * loopProb + (1 - loopProb) * pi


def elbo_value(log_pX, alpha, invL, Fb):
    """ref: VBx.py:100 (eq 25)."""
    return log_pX + Fb * 0.5 * np.sum(np.log(invL) - invL - alpha ** 2 + 1)


def prior_update(gamma, pi, loopProb, lfw, lbw, log_p, log_pX):
    """New speaker priors.  ref: VBx.py:101-104 (eq 24)."""
    entered = np.exp(logsumexp(lfw[:-1], axis=1, keepdims=True) + log_p[1:] + lbw[1:] - log_pX)
    new_pi = gamma[0] + (1 - loopProb) * pi * np.sum(entered, axis=0)
    return new_pi / new_pi.sum()


# --------------------------------------------------------------------------------------
# the VB loop                                                       ref: VBx.py:27-126
# --------------------------------------------------------------------------------------
def VBx(X, Phi, loopProb=0.9, Fa=1.0, Fb=1.0, pi=10, gamma=None, maxIters=10,
        epsilon=1e-4, alphaQInit=1.0, ref=None, plot=False,
        return_model=False, alpha=None, invL=None):
    """Same call signature and return values as the reference (VBx.py:27-29, :126).
    ``plot`` is accepted and ignored (the oracle never draws)."""
    if type(pi) is int:                                        # VBx.py:76-77 (Python int only)
        pi = np.ones(pi) / pi
    if gamma is None:                                          # VBx.py:79-83 (global NumPy RNG)
        gamma = np.random.gamma(alphaQInit, size=(X.shape[0], len(pi)))
        gamma = gamma / gamma.sum(1, keepdims=True)
    assert gamma.shape[1] == len(pi) and gamma.shape[0] == X.shape[0]   # VBx.py:85

    G, rho = frame_constants(X, Phi)
    history = []
    for it in range(maxIters):
        if it > 0 or alpha is None or invL is None:            # VBx.py:94
            alpha, invL = speaker_model(gamma, rho, Phi, Fa, Fb)
        log_p = frame_loglik(rho, alpha, invL, Phi, G, Fa)
        tr = transition_matrix(pi, loopProb)
        gamma, log_pX, lfw, lbw = forward_backward(log_p, tr, pi)
        elbo = elbo_value(log_pX, alpha, invL, Fb)
        pi = prior_update(gamma, pi, loopProb, lfw, lbw, log_p, log_pX)
        history.append([elbo])
        if ref is not None:                                    # VBx.py:108-109
            history[-1] += [DER(gamma, ref), DER(gamma, ref, xentropy=True)]
        if it > 0 and elbo - history[-2][0] < epsilon:         # VBx.py:122-125
            if elbo - history[-2][0] < 0:
                print('WARNING: Value of auxiliary function has decreased!')
            break
    out = (gamma, pi, history)
    if return_model:
        out = out + (alpha, invL)
    return out


def DER(q, ref, expected=True, xentropy=False):
    """Expected / hard diarization error or frame cross-entropy under the best
    speaker permutation.  ref: VBx.py:134-143."""
    from scipy.optimize import linear_sum_assignment
    from scipy.sparse import coo_matrix
    n = len(ref)
    if not expected:
        q = coo_matrix((np.ones(len(q)), (range(len(q)), q.argmax(1)))).toarray()
    ref_onehot = coo_matrix((np.ones(n), (range(n), ref)))
    cost = ref_onehot.T.dot(-np.log(q + np.nextafter(0, 1)) if xentropy else -q)
    best = cost[linear_sum_assignment(cost)].sum()
    return best / float(n) if xentropy else (n + best) / float(n)


# --------------------------------------------------------------------------------------
# scaled linear-domain restatement
# --------------------------------------------------------------------------------------
def fb_linear(lls, pi, loopProb):
    """Forward-backward for the transition matrix of VBx.py:98 without forming it.

    With c_j = (1-loopProb)*pi_j + 1e-8 the matrix (tr + 1e-8) of VBx.py:158 equals
    loopProb*I + 1 c^T, so one step costs O(S):

        a_t[j]  = B_t[j] * (loopProb * a_{t-1}[j] + c_j * sum_i a_{t-1}[i])       (VBx.py:167-168)
        be_t[i] = loopProb * e_{t+1}[i] + sum_j c_j e_{t+1}[j],  e = B * be        (VBx.py:170-171)

    Everything is kept normalised (row-max shifted likelihoods, forward vectors that
    sum to one); the total log-likelihood is the sum of the log scales.  Returns
    ``(gamma, tll, entered)`` where ``entered[j] = sum_{t>=1} exp(LSE_i lfw[t-1,i] +
    lls[t,j] + lbw[t,j] - tll)`` is the statistic of VBx.py:101-103, which in this
    domain is ``gamma[t,j] / (loopProb*ahat[t-1,j] + c_j)`` (App. A.4).
    """
    lls = np.asarray(lls, dtype=np.float64)
    T, S = lls.shape
    c = (1.0 - loopProb) * np.asarray(pi, dtype=np.float64) + EPS_TR
    m = lls.max(axis=1)
    B = np.exp(lls - m[:, None])
    ahat = np.empty((T, S))
    log_scale = np.empty(T)
    a = B[0] * (np.asarray(pi, dtype=np.float64) + EPS_TR)     # VBx.py:163
    s = a.sum()
    ahat[0] = a / s
    log_scale[0] = np.log(s)
    for t in range(1, T):
        a = B[t] * (loopProb * ahat[t - 1] + c)                # sum(ahat[t-1]) == 1
        s = a.sum()
        ahat[t] = a / s
        log_scale[t] = np.log(s)
    tll = float(log_scale.sum() + m.sum())
    bhat = np.empty((T, S))
    bhat[T - 1] = 1.0
    for t in range(T - 2, -1, -1):
        e = B[t + 1] * bhat[t + 1]
        nb = loopProb * e + np.dot(c, e)
        bhat[t] = nb / nb.max()                                # any positive scale is fine
    g = ahat * bhat
    gamma = g / g.sum(axis=1, keepdims=True)
    entered = np.zeros(S)
    if T > 1:
        entered = (gamma[1:] / (loopProb * ahat[:-1] + c)).sum(axis=0)
    return gamma, tll, entered


def vb_iteration_linear(rho, Phi, gsum, gamma, pi, loopProb, Fa, Fb, alpha=None, invL=None):
    """One VB iteration (VBx.py:94-104) on top of ``fb_linear``.  ``gsum`` is
    sum_t G_t (VBx.py:87); G is left out of the T x S matrix because it cancels in
    gamma and pi and only shifts the total log-likelihood by Fa*gsum (App. A.5)."""
    if alpha is None or invL is None:
        alpha, invL = speaker_model(gamma, rho, Phi, Fa, Fb)
    lls = Fa * (rho.dot(alpha.T) - 0.5 * (invL + alpha ** 2).dot(Phi))
    new_gamma, tll, entered = fb_linear(lls, pi, loopProb)
    log_pX = tll + Fa * gsum
    elbo = elbo_value(log_pX, alpha, invL, Fb)
    new_pi = new_gamma[0] + (1 - loopProb) * pi * entered
    new_pi = new_pi / new_pi.sum()
    return new_gamma, new_pi, elbo, alpha, invL
