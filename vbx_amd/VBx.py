"""Host-side mirror of the reference module ``VBx/VBx.py`` on top of libvbx_hip.so.

Same public names, call signatures, return values and error behaviour as the reference
(``VBx`` VBx.py:27-126, ``forward_backward`` VBx.py:146-175, ``DER`` VBx.py:134-143), so the
reference driver ``vbhmm.py`` (``from VBx import VBx``, vbhmm.py:45, call site :154-158)
can import this module unchanged -- see ``vbx_drop_in/VBx.py`` and INTEGRATION.md.

All numerical work of the VB loop runs in HIP kernels on an MI355X; this file only marshals
arguments (dtype/contiguity, the ``pi`` int -> uniform expansion, the global-RNG gamma
initialisation) and replays the reference's observable host behaviour (the WARNING print,
``Li`` as a list of lists, DER bookkeeping when ``ref`` is given).  There is no CPU fallback:
without the library or a gfx950 device the call raises ``vbx_amd._capi.VbxError``.

Two extra keyword-only arguments (the reference call sites never pass them):
  precision  'fp64' | 'fp32' | None.  None -> $VBX_AMD_PRECISION or "auto": float32 ``X``
             selects the fp32 device path, anything else the fp64 path (the reference always
             computes in float64, VBx.py:87).  The fp64 path reproduces the reference's
             iteration count under ``epsilon``; fp32 matches gamma/pi/Li to ~1e-5 relative.
  device     HIP device index (default $VBX_AMD_DEVICE / $LOCAL_RANK / 0).
"""
from __future__ import annotations

import os

import numpy as np

from . import _capi

__all__ = ['VBx', 'forward_backward', 'DER']

_EPS_TR = 1e-8          # VBx.py:158


def _pick_precision(precision, X):
    if precision is None:
        precision = os.environ.get('VBX_AMD_PRECISION', 'auto')
    if precision == 'auto':
        return 'fp32' if getattr(X, 'dtype', None) == np.float32 else 'fp64'
    return precision


def VBx(X, Phi, loopProb=0.9, Fa=1.0, Fb=1.0, pi=10, gamma=None, maxIters=10,
        epsilon=1e-4, alphaQInit=1.0, ref=None, plot=False,
        return_model=False, alpha=None, invL=None, *, precision=None, device=None):
    """Variational-Bayes HMM over an x-vector sequence; see the reference docstring
    (VBx.py:30-67) for the meaning of every argument.  Returns ``(gamma[T,S], pi[S], Li)``
    plus ``(alpha[S,D], invL[S,D])`` when ``return_model`` is set."""
    X = np.asarray(X)
    Phi = np.asarray(Phi)
    if type(pi) is int:                                   # VBx.py:76-77: Python int only
        pi = np.ones(pi) / pi
    if gamma is None:                                     # VBx.py:79-83: global NumPy RNG
        gamma = np.random.gamma(alphaQInit, size=(X.shape[0], len(pi)))
        gamma = gamma / gamma.sum(1, keepdims=True)
    assert (gamma.shape[1] == len(pi) and gamma.shape[0] == X.shape[0])   # VBx.py:85

    pi = np.array(pi, dtype=np.float64)
    T, D = X.shape
    S = len(pi)
    if maxIters <= 0:                                     # range(0): inputs come back untouched
        return (gamma, pi, []) + ((alpha, invL) if return_model else ())

    ctx = _capi.default_context(device)
    batch = _capi.Batch(ctx, [T], [S], D, precision=_pick_precision(precision, X), max_iters=int(maxIters))
    try:
        batch.set_recording(0, X, Phi, pi, gamma, loopProb, Fa, Fb, alpha0=alpha, invL0=invL)
        Li = []
        if ref is None:
            batch.run(int(maxIters), epsilon)
            res = batch.result(0, want_model=return_model)
            Li = [[np.float64(e)] for e in res['Li']]
        else:
            # DER / cross-entropy per iteration need gamma on the host every iteration (VBx.py:108-120)
            batch.set_option(_capi.OPT_CHECK_EVERY, 1)
            res = None
            for ii in range(int(maxIters)):
                batch.run(1, epsilon)
                res = batch.result(0, want_model=return_model)
                if len(res['Li']) == len(Li):             # frozen: converged in the previous pass
                    break
                Li.append([np.float64(res['Li'][-1]), DER(res['gamma'], ref),
                           DER(res['gamma'], ref, xentropy=True)])
                if plot:
                    _plot_iteration(res['gamma'], ref, ii, maxIters)
                if ii > 0 and Li[-1][0] - Li[-2][0] < epsilon:
                    break
        if res['warned']:
            print('WARNING: Value of auxiliary function has decreased!')   # VBx.py:123-124
    finally:
        batch.close()
    out = (res['gamma'], res['pi'], Li)
    if return_model:
        out = out + (res['alpha'], res['invL'])
    return out


def _plot_iteration(gamma, ref, ii, maxIters):            # VBx.py:111-120
    import matplotlib.pyplot as plt
    if ii == 0:
        plt.clf()
    plt.subplot(maxIters, 1, ii + 1)
    plt.plot(gamma, lw=2)
    plt.imshow(np.atleast_2d(ref), interpolation='none', aspect='auto', cmap=plt.cm.Pastel1,
               extent=(0, len(ref), -0.05, 1.05))


def DER(q, ref, expected=True, xentropy=False):
    """Diarization error rate (or frame cross-entropy) between per-frame speaker posteriors
    ``q[T,S]`` and reference labels ``ref[T]`` under the best one-to-one speaker mapping.
    Host-side by design (Hungarian assignment); same semantics as VBx.py:134-143."""
    from scipy.optimize import linear_sum_assignment
    from scipy.sparse import coo_matrix
    q = np.asarray(q)
    n_frames = len(ref)
    if not expected:                                      # harden q to one-hot decisions
        hard = np.zeros_like(q, dtype=np.float64)
        hard[np.arange(len(q)), q.argmax(1)] = 1.0
        q = hard
    membership = coo_matrix((np.ones(n_frames), (np.arange(n_frames), ref)))
    per_pair = membership.T.dot(-np.log(q + np.nextafter(0, 1)) if xentropy else -q)
    rows, cols = linear_sum_assignment(per_pair)
    best = per_pair[rows, cols].sum()
    if xentropy:
        return best / float(n_frames)
    return (n_frames + best) / float(n_frames)


def _split_transition(tr):
    """tr = loopProb*I + 1 (x) off  ->  (loopProb, off[S]), or None if tr is not of that form (VBx.py:98 builds
    exactly this form; the kernels of the EM loop exploit it)."""
    S = tr.shape[0]
    if S == 1:
        return 0.0, tr[0].copy()
    off = np.where(np.eye(S, dtype=bool), np.nan, tr)
    col = np.nanmean(off, axis=0)
    if not np.allclose(np.nan_to_num(off - col), 0.0, rtol=0, atol=1e-12 * max(1.0, np.abs(tr).max())):
        return None
    diag = np.diag(tr) - col
    if not np.allclose(diag, diag[0], rtol=0, atol=1e-12):
        return None
    lp = float(diag[0])
    if not 0.0 <= lp <= 1.0 or (lp == 1.0 and np.any(col != 0.0)):
        return None
    return lp, col


def forward_backward(lls, tr, ip, *, precision='fp64', device=None):
    """HMM state posteriors.  Same signature and return tuple as VBx.py:146-175: ``(post[T,S], tll, lfw[T,S],
    lbw[T,S])``, for any transition matrix: the structure VBx() itself builds (``I*loopProb + (1-loopProb)*pi``,
    VBx.py:98) runs on the kernels of the EM loop, everything else on the dense kernel (vbx_fb_dense.hpp)."""
    lls = np.asarray(lls, dtype=np.float64)
    ip = np.asarray(ip, dtype=np.float64)
    tr = np.asarray(tr, dtype=np.float64)
    if lls.ndim != 2 or tr.shape != (lls.shape[1], lls.shape[1]) or ip.shape != (lls.shape[1],):
        raise ValueError('forward_backward: lls [T][S], tr [S][S], ip [S] expected')
    ctx = _capi.default_context(device)
    split = _split_transition(tr)
    if split is None:
        post, tll, lfw, lbw = ctx.forward_backward_dense(lls, tr, ip, precision=precision)
        return post, np.float64(tll), lfw, lbw
    lp, off = split
    pi_eff = off / (1.0 - lp) if lp < 1.0 else np.zeros_like(off)
    post, tll, _entered, lfw, lbw = ctx.forward_backward(lls, pi_eff, lp, ip=ip, precision=precision,
                                                         want_logs=True)
    return post, np.float64(tll), lfw, lbw
