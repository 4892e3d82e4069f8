"""Many independent recordings: one batch per GPU, recordings sharded across ranks.

The reference processes recordings one after another in a Python loop (vbhmm.py:117-123);
they share nothing but the read-only PLDA, so the path shards embarrassingly:
recording b goes to rank ``assignment[b]``, every rank runs its shard through one
``vbx_batch`` on its own GPU, and the only communication is a gather of the (small) results /
timings at the end -- no collective on the data path.
"""
from __future__ import annotations

import numpy as np

__all__ = ['shard_recordings', 'VBx_batch', 'VBx_batch_distributed']


def shard_recordings(costs, world_size: int):
    """Longest-processing-time-first assignment of recordings to ranks.

    ``costs[b]`` ~ T_b * S_b.  Returns ``assignment`` (rank per recording); deterministic, so
    every rank computes the same table without talking to the others."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind='stable')
    load = np.zeros(world_size)
    assignment = np.empty(len(costs), dtype=np.int64)
    for b in order:
        r = int(np.argmin(load))          # ties -> lowest rank
        assignment[b] = r
        load[r] += costs[b]
    return assignment


def _normalise(rec, defaults):
    """rec: dict with X, Phi and optional VBx() keyword arguments -> full argument dict."""
    kw = dict(loopProb=0.9, Fa=1.0, Fb=1.0, pi=10, gamma=None, alphaQInit=1.0, alpha=None, invL=None)
    for where in (defaults, rec):
        bad = set(where) - set(kw) - {'X', 'Phi'}
        if bad:            # maxIters / epsilon are per batch; anything else is a typo that would be ignored silently
            raise TypeError(f'VBx_batch: unexpected per-recording argument(s) {sorted(bad)} (maxIters and epsilon '
                            'apply to the whole batch)')
    kw.update(defaults)
    kw.update({k: v for k, v in rec.items() if k not in ('X', 'Phi')})
    X = np.asarray(rec['X'])
    pi = kw['pi']
    if type(pi) is int:                                   # VBx.py:76-77
        pi = np.ones(pi) / pi
    pi = np.array(pi, dtype=np.float64)
    gamma = kw['gamma']
    if gamma is None:                                     # VBx.py:79-83 (global RNG, in list order)
        gamma = np.random.gamma(kw['alphaQInit'], size=(X.shape[0], len(pi)))
        gamma = gamma / gamma.sum(1, keepdims=True)
    assert gamma.shape[1] == len(pi) and gamma.shape[0] == X.shape[0]     # VBx.py:85
    return dict(X=X, Phi=np.asarray(rec['Phi']), pi=pi, gamma=gamma, loopProb=kw['loopProb'], Fa=kw['Fa'],
                Fb=kw['Fb'], alpha=kw['alpha'], invL=kw['invL'])


def run_shard_hip(items, maxIters, epsilon, precision=None, device=None):
    """Run normalised recordings on the local GPU, one vbx_batch per feature dimension.  ``precision=None`` is
    VBx()'s rule (VBX_AMD_PRECISION, else fp32 only when every X of the batch is float32)."""
    from . import _capi
    from .VBx import _pick_precision
    ctx = _capi.default_context(device)
    results = [None] * len(items)
    by_dim = {}
    for k, it in enumerate(items):
        by_dim.setdefault(it['X'].shape[1], []).append(k)
    for D, idx in by_dim.items():
        prec = {_pick_precision(precision, items[k]['X']) for k in idx}
        batch = _capi.Batch(ctx, [items[k]['X'].shape[0] for k in idx], [len(items[k]['pi']) for k in idx], D,
                            precision='fp64' if 'fp64' in prec else prec.pop(), max_iters=maxIters)
        try:
            for j, k in enumerate(idx):
                it = items[k]
                batch.set_recording(j, it['X'], it['Phi'], it['pi'], it['gamma'], it['loopProb'], it['Fa'],
                                    it['Fb'], alpha0=it['alpha'], invL0=it['invL'])
            batch.run(maxIters, epsilon)
            for j, k in enumerate(idx):
                results[k] = batch.result(j)
        finally:
            batch.close()
    return results


def VBx_batch(recordings, maxIters=10, epsilon=1e-4, precision=None, device=None, return_model=False,
              **defaults):
    """``[VBx(**rec, maxIters=..., epsilon=...) for rec in recordings]`` on one GPU, in one batch.

    Each recording is a dict with ``X`` and ``Phi`` plus any of VBx()'s keyword arguments;
    ``defaults`` supplies shared hyper-parameters; ``maxIters`` and ``epsilon`` apply to the whole batch.  ``precision``
    follows VBx(): fp64 kernels unless every X is float32 (or VBX_AMD_PRECISION / the argument says otherwise), so the
    numerics and iteration counts are those of one VBx() call per recording.  Returns a list of ``(gamma, pi, Li[,
    alpha, invL])`` tuples in input order (same types as the reference returns, VBx.py:126)."""
    items = [_normalise(r, defaults) for r in recordings]
    if maxIters <= 0:
        return [(it['gamma'], it['pi'], []) + ((it['alpha'], it['invL']) if return_model else ())
                for it in items]
    raw = run_shard_hip(items, int(maxIters), epsilon, precision=precision, device=device)
    return [_as_tuple(r, return_model) for r in raw]


def _as_tuple(res, return_model):
    if res['warned']:
        print('WARNING: Value of auxiliary function has decreased!')       # VBx.py:123-124
    out = (res['gamma'], res['pi'], [[np.float64(e)] for e in res['Li']])
    if return_model:
        out = out + (res['alpha'], res['invL'])
    return out


def VBx_batch_distributed(recordings, maxIters=10, epsilon=1e-4, precision=None, return_model=False,
                          run_shard=None, gather=True, **defaults):
    """Shard ``recordings`` over the ranks of the initialised ``torch.distributed`` group.

    Every rank passes the same list; rank r computes the recordings assigned to it and, with
    ``gather=True``, an ``all_gather_object`` hands every rank the complete result list (the
    results are a few MB at most: no data-path collective).  ``run_shard(items, maxIters,
    epsilon)`` defaults to the HIP path; the CPU test-suite injects the oracle here."""
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    items = [_normalise(r, defaults) for r in recordings]                  # same RNG draws on every rank
    costs = [it['X'].shape[0] * len(it['pi']) for it in items]
    assignment = shard_recordings(costs, world)
    mine = [b for b in range(len(items)) if assignment[b] == rank]
    if run_shard is None:
        def run_shard(sub, mi, eps):
            return run_shard_hip(sub, mi, eps, precision=precision)
    local = run_shard([items[b] for b in mine], int(maxIters), epsilon) if mine else []
    local = {b: res for b, res in zip(mine, local)}
    if not gather or world == 1:
        merged = local
    else:
        parts = [None] * world
        dist.all_gather_object(parts, local)
        merged = {}
        for p in parts:
            merged.update(p)
    return [(_as_tuple(merged[b], return_model) if b in merged else None) for b in range(len(items))]
