"""NumPy model of the chunked parallel scan (vbx_amd/csrc/vbx_scan.hpp, vbx_operator.hpp, the re-run of
vbx_chunk_post.hpp)  --  TEST INFRASTRUCTURE.

Mirrors the three device kernels step for step (power-of-two column rescaling with integer
exponents, support-aware weighting in the boundary chain, per-frame normalisation in the
re-run) so that the algebra, including its corner cases (exact zeros in b, zero operator
columns, one-frame chunks), can be checked on the CPU against ``vbx_oracle.fb_linear`` and the
reference's log-domain answers.  ``dtype`` selects the working precision (float32 mimics the
fp32 device path).
"""
from __future__ import annotations

import numpy as np

EPS = 1e-8
NEG_BIG = -10 ** 9


def _rescale_exponent(sig, dtype, clamp=True):
    lim = (126 if dtype is np.float32 else 1022) if clamp else 10 ** 6
    return np.where(sig > 0, np.clip(np.frexp(sig)[1], -lim, lim), 0).astype(np.int64)


def scan1(B, c, lp, dtype, first_chunk, direction, zero_column_fix=True, clamp=True):
    """Transfer operator of one chunk.  Returns (cols[S,S] with cols[i] = operator column i, expo[S])."""
    L, S = B.shape
    X = np.eye(S, dtype=dtype)                    # X[i] = column i (a vector over states)
    expo = np.zeros(S, dtype=np.int64)
    lp = dtype(lp)
    e_prev = np.zeros(S, dtype=np.int64)
    for step in range(L):
        if direction == 0:
            sig = X.sum(axis=1)
            e = _rescale_exponent(sig, dtype, clamp)
        else:
            e = e_prev                                # backward: exponent of the previous frame's q
        expo += e
        if direction == 0:
            b = B[step]
            first = first_chunk and step == 0
            lps = np.ldexp(dtype(1) if first else lp, -e).astype(dtype)
            sgs = np.zeros(S, dtype=dtype) if first else np.ldexp(sig, -e).astype(dtype)
            X = (b[None, :] * (lps[:, None] * X + c[None, :] * sgs[:, None])).astype(dtype)
        else:
            b = B[L - 1 - step]
            sc = np.ldexp(dtype(1), -e).astype(dtype)
            U = (b[None, :] * (X * sc[:, None])).astype(dtype)
            q = (U * c[None, :]).sum(axis=1).astype(dtype)
            X = (lp * U + q[:, None]).astype(dtype)
            e_prev = _rescale_exponent(q, dtype, clamp)
    sig = X.sum(axis=1)
    e = _rescale_exponent(sig, dtype, clamp)
    expo += e
    X = np.ldexp(X, -e[:, None]).astype(dtype)
    if zero_column_fix:
        expo = np.where(sig > 0, expo, NEG_BIG)   # a zero column carries no weight at all
    return X, expo


def scan2_apply_transposed(g, cols, expo, dtype):
    """Backward boundary step: B_k = F_k^T, so (B g)_i = 2^{E_i} <col_i, g>; outputs rescaled by the
    largest exponent present."""
    dots = (cols * g[None, :]).sum(axis=1).astype(dtype)
    pos = dots > 0
    with np.errstate(divide='ignore'):
        tj = np.where(pos, expo + np.frexp(dots)[1], NEG_BIG * 4)
    top = tj.max()
    return np.where(pos, np.ldexp(dots, np.clip(expo - top, -100000, 100000)), 0).astype(dtype)


def scan2_apply(y, cols, expo, dtype):
    pos = y > 0
    with np.errstate(divide='ignore'):
        tj = np.where(pos, expo + np.frexp(y)[1], NEG_BIG * 4)
    top = tj.max()
    w = np.where(pos, np.ldexp(y, np.clip(expo - top, -100000, 100000)), 0).astype(dtype)
    return (w[:, None] * cols).sum(axis=0).astype(dtype)

def compose(op_next, op_prev, dtype):
    """Operator of two consecutive chunk ranges: (cols, expo) of F_next F_prev (scan_compose kernel).
    Every column of F_prev is pushed through F_next exactly like a boundary vector in scan2_apply, but its
    scale is kept: column i of the product = 2^(expo_prev[i] + top_i) * sum_j w_j col_next_j, renormalised so
    that the column sum lies in [0.5, 1)."""
    cols_n, expo_n = op_next
    cols_p, expo_p = op_prev
    S = cols_p.shape[0]
    out = np.zeros_like(cols_p)
    eo = np.full(S, NEG_BIG, dtype=np.int64)
    for i in range(S):
        y = cols_p[i]
        pos = y > 0
        if not pos.any() or expo_p[i] <= NEG_BIG // 2:
            continue
        with np.errstate(divide='ignore'):
            tj = np.where(pos, expo_n + np.frexp(y)[1], NEG_BIG * 4)
        top = tj.max()
        if top <= NEG_BIG:                      # the support of the column meets only zero columns of F_next
            continue
        w = np.where(pos, np.ldexp(y, np.clip(expo_n - top, -100000, 100000)), 0).astype(dtype)
        v = (w[:, None] * cols_n).sum(axis=0).astype(dtype)
        sig = v.sum(dtype=dtype)
        if not sig > 0:
            continue
        e = int(_rescale_exponent(np.asarray(sig), dtype))
        out[i] = np.ldexp(v, -e).astype(dtype)
        eo[i] = expo_p[i] + top + e
    return out, eo


def forward_backward_chunked(lls, pi, loopProb, ip=None, chunk=128, dtype=np.float64, pad_to=None,
                             zero_column_fix=True, clamp=True, super_group=1, split_halves=False):
    """gamma, tll, entered -- same contract as vbx_oracle.fb_linear, computed the chunked way.
    ``pad_to`` appends padded speakers exactly as the device layout does (b = 0, c = 0, no initial
    mass); ``zero_column_fix=False`` reproduces the bug the first device version had.  ``super_group`` > 1 walks
    the chunk boundaries in two levels (compose groups of that many chunk operators, walk the groups, then walk
    inside every group), as the device does for long recordings.  ``split_halves`` models the fused kernels of round 2:
    every chunk longer than half the chunk length gets the operators of its two halves (chunk_loglik), the chunk operator
    of the boundary walk is their composition, and the re-run (chunk_post) starts the second half's forward recursion
    from P1 applied to the chunk's forward boundary and the first half's backward recursion from P2^T applied to its
    backward boundary."""
    lls = np.asarray(lls, dtype=np.float64)
    T, S_true = lls.shape
    pi = np.asarray(pi, dtype=np.float64)
    ip = pi if ip is None else np.asarray(ip, dtype=np.float64)
    m = lls.max(axis=1).astype(dtype)
    B = np.exp(lls - m[:, None].astype(np.float64)).astype(dtype)
    c = ((1.0 - loopProb) * pi + EPS).astype(dtype)
    ip0 = (ip + EPS).astype(dtype)
    S = S_true
    if pad_to is not None and pad_to > S_true:
        S = pad_to
        B = np.concatenate([B, np.zeros((T, S - S_true), dtype=dtype)], axis=1)
        c = np.concatenate([c, np.zeros(S - S_true, dtype=dtype)])
        ip0 = np.concatenate([ip0, np.zeros(S - S_true, dtype=dtype)])
    lp = dtype(loopProb)
    starts = list(range(0, T, chunk))
    K = len(starts)
    ops = []
    halves = {}
    H = chunk // 2
    for k, t0 in enumerate(starts):
        Bk = B[t0:t0 + chunk]
        if split_halves and len(Bk) > H:
            p1 = scan1(Bk[:H], c, lp, dtype, k == 0, 0, zero_column_fix, clamp)
            p2 = scan1(Bk[H:], c, lp, dtype, False, 0, zero_column_fix, clamp)
            halves[k] = (p1, p2)
            ops.append(compose(p2, p1, dtype))
        else:
            ops.append(scan1(Bk, c, lp, dtype, k == 0, 0, zero_column_fix, clamp))
    fbound = [None] * K
    gbound = [None] * K
    fbound[0] = ip0
    gbound[K - 1] = np.concatenate([np.ones(S_true, dtype=dtype), np.zeros(S - S_true, dtype=dtype)])
    live = np.arange(S) < S_true
    def mask(v):
        return np.where(live, v, 0).astype(dtype) if zero_column_fix else v

    if super_group <= 1:
        for k in range(K - 1):
            fbound[k + 1] = scan2_apply(fbound[k], *ops[k], dtype)
        for k in range(K - 1, 0, -1):
            gbound[k - 1] = mask(scan2_apply_transposed(gbound[k], *ops[k], dtype))
    else:
        G = super_group
        starts_s = list(range(0, K, G))
        sops = []
        for a in starts_s:                            # level 1: one operator per group of G chunks
            b = min(a + G, K)
            acc = ops[a]
            for k in range(a + 1, b):
                acc = compose(ops[k], acc, dtype)
            sops.append(acc)
        for s_i, a in enumerate(starts_s[:-1]):       # level 2: boundaries at the group starts / ends
            fbound[starts_s[s_i + 1]] = scan2_apply(fbound[a], *sops[s_i], dtype)
        for s_i in range(len(starts_s) - 1, 0, -1):
            a = starts_s[s_i]
            b = min(a + G, K)
            gbound[a - 1] = mask(scan2_apply_transposed(gbound[b - 1], *sops[s_i], dtype))
        for a in starts_s:                            # level 3: inside every group
            b = min(a + G, K)
            for k in range(a, b - 1):
                fbound[k + 1] = scan2_apply(fbound[k], *ops[k], dtype)
            for k in range(b - 1, a, -1):
                gbound[k - 1] = mask(scan2_apply_transposed(gbound[k], *ops[k], dtype))
    ahat = np.empty((T, S), dtype=dtype)
    bhat = np.empty((T, S), dtype=dtype)
    tll = 0.0

    def rerun(t0, Bs, fvec, first, gvec):
        """Frames t0 .. t0+len(Bs)-1 from the forward vector entering them and the backward vector at their last frame."""
        nonlocal tll
        L = len(Bs)
        x = fvec
        if not first:
            x = (x / x.sum()).astype(dtype)
        for i in range(L):
            pre = x if (first and i == 0) else (lp * x + c).astype(dtype)
            u = (Bs[i] * pre).astype(dtype)
            r = u.sum(dtype=dtype)
            x = (u / r).astype(dtype)
            ahat[t0 + i] = x
            tll += float(np.log(np.float64(r))) + float(m[t0 + i])
        x = gvec
        with np.errstate(invalid='ignore', divide='ignore'):
            x = (x / x.sum() * S).astype(dtype)
        bhat[t0 + L - 1] = x
        for i in range(L - 1):
            u = (Bs[L - 1 - i] * x).astype(dtype)
            r = (c * u).sum(dtype=dtype)
            x = (u * (lp / r) + dtype(1)).astype(dtype)
            bhat[t0 + L - 2 - i] = x

    for k, t0 in enumerate(starts):
        Bk = B[t0:t0 + chunk]
        if k in halves:
            p1, p2 = halves[k]
            a_cut = scan2_apply(fbound[k], *p1, dtype)                       # a at frame H-1 (any scale)
            x_cut = mask(scan2_apply_transposed(gbound[k], *p2, dtype))      # x at frame H-1 (any scale)
            rerun(t0, Bk[:H], fbound[k], k == 0, x_cut)
            rerun(t0 + H, Bk[H:], a_cut, False, gbound[k])
        else:
            rerun(t0, Bk, fbound[k], k == 0, gbound[k])
    g = ahat.astype(np.float64) * bhat.astype(np.float64)
    with np.errstate(invalid='ignore', divide='ignore'):
        gamma = g / g.sum(axis=1, keepdims=True)
        entered = np.zeros(S)
        if T > 1:
            den = loopProb * ahat[:-1].astype(np.float64) + c.astype(np.float64)
            entered = np.where(live, (gamma[1:] / np.where(den > 0, den, 1.0)).sum(axis=0), 0.0)
    return gamma[:, :S_true], tll, entered[:S_true]
