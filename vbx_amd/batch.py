"""Many independent recordings: one batch per GPU, recordings sharded across ranks.

The reference processes recordings one after another in a Python loop (vbhmm.py:117-123);
they share nothing but the read-only PLDA, so the path shards embarrassingly:
recording b goes to rank ``assignment[b]``, every rank runs its shard through one
``vbx_batch`` on its own GPU, and the only communication is a gather of the (small) results /
timings at the end -- no collective on the data path.
"""
from __future__ import annotations

import os

import numpy as np

__all__ = ['shard_recordings', 'VBx_batch', 'VBx_sweep', 'VBx_batch_distributed']


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


# VBx() keyword arguments that mean something per recording, and the ones that only exist per batch / per call: the
# latter are accepted in a recording's dict (a dict built for VBx(**kw) can be passed as it is) and ignored with a warning
# when they differ from what the batch runs with
PER_RECORDING = ('loopProb', 'Fa', 'Fb', 'pi', 'gamma', 'alphaQInit', 'alpha', 'invL')
PER_BATCH = ('maxIters', 'epsilon', 'return_model', 'ref', 'plot', 'precision', 'device')


def _normalise(rec, defaults):
    """rec: dict with X, Phi and optional VBx() keyword arguments -> full argument dict."""
    kw = dict(loopProb=0.9, Fa=1.0, Fb=1.0, pi=10, gamma=None, alphaQInit=1.0, alpha=None, invL=None)
    for where in (defaults, rec):
        bad = set(where) - set(PER_RECORDING) - set(PER_BATCH) - {'X', 'Phi'}
        if bad:            # a typo would otherwise be ignored silently
            raise TypeError(f'VBx_batch: unexpected per-recording argument(s) {sorted(bad)}; accepted: '
                            f'{", ".join(PER_RECORDING)} (and, ignored, the per-batch ones: {", ".join(PER_BATCH)})')
        ignored = sorted(k for k in where if k in PER_BATCH and where[k] is not None and where[k] is not False)
        if ignored and where is rec:
            import warnings
            warnings.warn(f'VBx_batch: {ignored} in a recording\'s arguments apply to the whole batch (pass them to '
                          'VBx_batch itself); ignored here', stacklevel=3)
    kw.update({k: v for k, v in defaults.items() if k in PER_RECORDING})
    kw.update({k: v for k, v in rec.items() if k in PER_RECORDING})
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


def _shape_of(rec, defaults):
    """(T, S, alphaQInit or None) of a recording WITHOUT normalising it: what sharding needs from a recording another rank
    will run.  The third entry is the Dirichlet parameter when the recording draws its initialisation from the global RNG
    (``gamma`` absent or None, VBx.py:79-83), else None.  Same validation of the keys as _normalise."""
    bad = (set(rec) | set(defaults)) - set(PER_RECORDING) - set(PER_BATCH) - {'X', 'Phi'}
    if bad:
        raise TypeError(f'VBx_batch: unexpected per-recording argument(s) {sorted(bad)}; accepted: '
                        f'{", ".join(PER_RECORDING)} (and, ignored, the per-batch ones: {", ".join(PER_BATCH)})')
    kw = {k: v for k, v in defaults.items() if k in PER_RECORDING}
    kw.update({k: v for k, v in rec.items() if k in PER_RECORDING})
    pi = kw.get('pi', 10)
    S = pi if type(pi) is int else len(pi)
    xs = np.shape(rec['X'])
    if len(xs) != 2:
        raise ValueError(f'VBx_batch: X must be T x D, got shape {xs}')
    T = xs[0]
    # the checks of _normalise that need no copy, on EVERY rank: a recording that cannot run fails everywhere at once,
    # before any rank enters a collective (round 6; owner-only validation left the other ranks waiting in the gather)
    if np.shape(rec['Phi']) != (xs[1],):
        raise ValueError(f'VBx_batch: Phi has shape {np.shape(rec["Phi"])}, X has {xs[1]} dimensions')
    if kw.get('gamma') is not None:
        assert np.shape(kw['gamma']) == (T, S), (np.shape(kw['gamma']), (T, S))     # VBx.py:85
    for name in ('alpha', 'invL'):
        if kw.get(name) is not None and np.shape(kw[name]) != (S, xs[1]):
            raise ValueError(f'VBx_batch: {name} has shape {np.shape(kw[name])}, expected {(S, xs[1])}')
    return int(T), int(S), (kw.get('alphaQInit', 1.0) if kw.get('gamma') is None else None)


def _padded_states(n_states):
    """The padded state count a vbx_batch of this many speakers runs with (vbx_host_batch.hpp: powers of two from 16)."""
    sp = 16
    while sp < n_states:
        sp *= 2
    return sp


def _pipeline_halves(n_rec, n_bytes):
    """Run a batch call as two halves on two contexts, the second half's uploads behind the first half's iterations and the
    first half's results behind the second half's iterations?  Only on request (VBX_AMD_BATCH_PIPELINE=1)."""
    # Measured (64 recordings of T = 10 000, 40 iterations): 27.4 -> 24.6 ms per call, fp64 45.6 -> 39.9, 128 recordings 42.1 ->
    # 37.4; 32 recordings 14.7 -> 15.5, 16: 8.6 -> 10.3 (the second context's threads and synchronisation cost a millisecond)
    # -- in a process that holds no other streams.  Inside bench.py (a ctx of its own, the default ctx and the second one: a
    # dozen streams on eight hardware queues, two busy streams to a queue) the same call takes 30.1 ms against 28.0 plain, so
    # the pipeline is NOT the default: VBX_AMD_BATCH_PIPELINE=1 asks for it (it pays from 48 recordings and 256 MB).
    return os.environ.get('VBX_AMD_BATCH_PIPELINE') == '1' and n_rec >= 2


def _run_one_batch(ctx, items, idx, D, prec, maxIters, epsilon, results, gates=None):
    """One vbx_batch for the recordings ``idx`` of ``items`` on ``ctx``: enqueue the uploads (one host thread per stream),
    iterate, fetch.  ``gates`` = (upload, run, fetch) locks shared with the other half of a pipelined call: the host link
    carries one half's arrays at a time in each direction, and at most one half iterates."""
    from . import _capi
    import contextlib
    up, run, fetch = gates if gates else (contextlib.nullcontext(),) * 3
    batch = _capi.Batch(ctx, [items[k]['X'].shape[0] for k in idx], [len(items[k]['pi']) for k in idx], D,
                        precision=prec, max_iters=maxIters)
    try:
        # uploads are only enqueued (one synchronize when the run begins, not one per recording) and the results of the
        # whole batch come back in one call into pinned host memory: what a batch call pays around its iterations
        batch.set_async_upload(True)

        def upload(pairs):
            for j, k in pairs:
                it = items[k]
                batch.set_recording(j, it['X'], it['Phi'], it['pi'], it['gamma'], it['loopProb'], it['Fa'],
                                    it['Fb'], alpha0=it['alpha'], invL0=it['invL'])
        by_stream = {}
        for j, k in enumerate(idx):
            by_stream.setdefault(batch.stream_of(j), []).append((j, k))
        with up:
            if len(by_stream) > 1:
                # one host thread per stream of the batch (the library releases the GIL in its calls; every stream has its own
                # staging block and device arena): the copies of the sub-batches share the host link instead of queueing up
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=len(by_stream)) as pool:
                    for f in [pool.submit(upload, pairs) for pairs in by_stream.values()]:
                        f.result()
            else:
                upload(list(enumerate(idx)))
            if gates:
                batch.sync_uploads()
        with run:
            batch.run(maxIters, epsilon)
        with fetch:
            for k, res in zip(idx, batch.results()):
                results[k] = res
    finally:
        batch.close()


def run_shard_hip(items, maxIters, epsilon, precision=None, device=None):
    """Run normalised recordings on the local GPU, one vbx_batch per (feature dimension, padded state count): every
    recording of a batch runs with the widest one's padding, and one recording with more than 64 speakers would push the
    others from the fused kernels onto the wide scan.  ``precision=None`` is VBx()'s rule (VBX_AMD_PRECISION, else fp32
    only when every X of the batch is float32).  With VBX_AMD_BATCH_PIPELINE=1 a batch runs as two halves on two contexts of
    the device: the second half uploads while the first iterates, the first half's results come back while the second
    iterates (_pipeline_halves)."""
    from . import _capi
    from .VBx import _pick_precision
    ctx = _capi.default_context(device)
    results = [None] * len(items)
    by_dim = {}
    for k, it in enumerate(items):
        by_dim.setdefault((it['X'].shape[1], _padded_states(len(it['pi']))), []).append(k)
    for (D, _sp), idx in by_dim.items():
        prec = {_pick_precision(precision, items[k]['X']) for k in idx}
        prec = 'fp64' if 'fp64' in prec else prec.pop()
        n_bytes = sum(items[k]['X'].nbytes + items[k]['gamma'].nbytes for k in idx)
        if _pipeline_halves(len(idx), n_bytes):
            import threading
            # halves of about equal cost (frames), in input order
            costs = np.cumsum([items[k]['X'].shape[0] for k in idx])
            cut = int(np.searchsorted(costs, costs[-1] / 2.0)) + 1
            halves = [idx[:cut], idx[cut:]] if 0 < cut < len(idx) else [idx]
            gates = (threading.Lock(), threading.Lock(), threading.Lock())
            errors = []

            def work(slot, part):
                try:
                    _run_one_batch(_capi.default_context(ctx.device, slot), items, part, D, prec, maxIters, epsilon, results, gates)
                except BaseException as exc:           # (re-raised on the calling thread)
                    errors.append(exc)
            threads = [threading.Thread(target=work, args=(slot, part)) for slot, part in enumerate(halves)]
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            if errors:
                raise errors[0]
        else:
            _run_one_batch(ctx, items, idx, D, prec, maxIters, epsilon, results)
    return results


def VBx_batch(recordings, maxIters=10, epsilon=1e-4, precision=None, device=None, return_model=False,
              **defaults):
    """``[VBx(**rec, maxIters=..., epsilon=...) for rec in recordings]`` on one GPU, in one batch.

    Each recording is a dict with ``X`` and ``Phi`` plus the per-recording keyword arguments of VBx(): ``loopProb``,
    ``Fa``, ``Fb``, ``pi``, ``gamma``, ``alphaQInit``, ``alpha``, ``invL``; ``defaults`` supplies shared values of the
    same.  ``maxIters``, ``epsilon``, ``return_model``, ``precision`` and ``device`` apply to the whole batch: given in a
    recording's dict they are ignored with a warning (``ref`` / ``plot`` need the responsibilities on the host every
    iteration and are a VBx() feature only); any other key is a TypeError.  ``precision`` follows VBx(): fp64 kernels
    unless every X is float32 (or VBX_AMD_PRECISION / the argument says otherwise) -- NB before round 2 the default was
    fp32 whatever the input type -- so the numerics and iteration counts are those of one VBx() call per recording.  Returns a list of ``(gamma, pi, Li[,
    alpha, invL])`` tuples in input order (same types as the reference returns, VBx.py:126)."""
    items = [_normalise(r, defaults) for r in recordings]
    if maxIters <= 0:
        return [(it['gamma'], it['pi'], []) + ((it['alpha'], it['invL']) if return_model else ())
                for it in items]
    raw = run_shard_hip(items, int(maxIters), epsilon, precision=precision, device=device)
    return [_as_tuple(r, return_model) for r in raw]


def sweep_streams(n_points, T):
    """HIP streams a sweep of ``n_points`` over one recording of T frames runs on.  Every stream's sub-batch keeps one copy of
    rho, shared by the points dealt to it; with two or three streams the latency-bound launches of one (boundary walk,
    per-recording reductions: 380 of 1490 us per iteration of BASELINE config 5 on one stream) hide behind the per-chunk
    kernels of the others.  ``VBX_AMD_SWEEP_STREAMS`` overrides; sweeps of short recordings stay on one stream (a launch
    must still fill the chip)."""
    import os
    env = os.environ.get('VBX_AMD_SWEEP_STREAMS')
    if env:
        return max(1, min(int(env), n_points, 8))
    tiles = n_points * ((T + 127) // 128)
    return 3 if (n_points >= 6 and tiles >= 4608) else 2 if (n_points >= 4 and tiles >= 3072) else 1


def run_sweep_hip(X, Phi, items, maxIters, epsilon, precision=None, device=None):
    """Normalised sweep points over ONE recording on the local GPU: one vbx_batch, the first point owns rho, the others
    share it (vbx_batch_set_recording_shared) -- one rho per stream sub-batch (sweep_streams)."""
    from . import _capi
    from .VBx import _pick_precision
    ctx = _capi.default_context(device)
    T, D = X.shape
    batch = _capi.Batch(ctx, [T] * len(items), [len(it['pi']) for it in items], D, precision=_pick_precision(precision, X),
                        max_iters=maxIters, streams=sweep_streams(len(items), T))
    try:
        for j, it in enumerate(items):
            if j == 0:
                batch.set_recording(0, X, Phi, it['pi'], it['gamma'], it['loopProb'], it['Fa'], it['Fb'],
                                    alpha0=it['alpha'], invL0=it['invL'])
            else:
                batch.set_recording_shared(j, 0, it['pi'], it['gamma'], it['loopProb'], it['Fa'], it['Fb'],
                                           alpha0=it['alpha'], invL0=it['invL'])
        batch.run(maxIters, epsilon)
        return [batch.result(j) for j in range(len(items))]
    finally:
        batch.close()


def VBx_sweep(X, Phi, points, maxIters=10, epsilon=1e-4, precision=None, device=None, return_model=False, **defaults):
    """``[VBx(X, Phi, **defaults, **p, maxIters=..., epsilon=...) for p in points]`` -- a hyper-parameter sweep over ONE
    recording, the grids of the reference's recipes (DIHARD2_run.sh:42-47, AMI_run.sh:44-49, CALLHOME_run.sh:42-47:
    Fa x Fb x loopP around one x-vector sequence) -- as one batch on one GPU with ONE rho = X * sqrt(Phi) (VBx.py:89) in
    HBM: the per-chunk kernels run the chunks of all points that read the same rows of it side by side, so HBM delivers
    the x-vectors once per kernel and not once per point.  ``points``: dicts of the per-recording keyword arguments of
    VBx() (``Fa``, ``Fb``, ``loopProb``, ``pi``, ``gamma``, ``alphaQInit``, ``alpha``, ``invL``).  ``gamma=None`` draws the
    initialisation from the global RNG per point, in list order, exactly as successive VBx() calls would.  Returns the
    list of ``(gamma, pi, Li[, alpha, invL])`` tuples."""
    X = np.asarray(X)
    items = [_normalise(dict(p, X=X, Phi=Phi), defaults) for p in points]
    if maxIters <= 0:
        return [(it['gamma'], it['pi'], []) + ((it['alpha'], it['invL']) if return_model else ()) for it in items]
    if not items:
        return []
    raw = run_sweep_hip(X, np.asarray(Phi), items, int(maxIters), epsilon, precision=precision, device=device)
    return [_as_tuple(r, return_model) for r in raw]


def _as_tuple(res, return_model, warn=True):
    if res['warned'] and warn:
        print('WARNING: Value of auxiliary function has decreased!')       # VBx.py:123-124
    out = (res['gamma'], res['pi'], [[np.float64(e)] for e in res['Li']])
    if return_model:
        out = out + (res['alpha'], res['invL'])
    return out


def VBx_batch_distributed(recordings, maxIters=10, epsilon=1e-4, precision=None, return_model=False,
                          run_shard=None, gather=True, gather_chunk_bytes=64 << 20, **defaults):
    """Shard ``recordings`` over the ranks of the initialised ``torch.distributed`` group.

    Every rank passes the same list; rank r computes the recordings assigned to it (LPT on T x S) on its own GPU.  There
    is no collective on the data path; what happens to the RESULTS afterwards is the caller's choice:

      gather=True / 'all'    ``all_gather_object``: every rank returns the complete list.  (Rounds 1-3 implemented ``True`` as
                             one gather to rank 0 while documenting "every rank"; since round 4 the code does what the
                             documentation said.  It costs world x the bytes of the responsibilities: 8 T S per recording
                             -- 2.4 MB at T = 10 000, S = 30, 154 MB per rank for BASELINE config 4 -- so anything that
                             only WRITES results should ask for 'root' or False)
      gather='root'          one ``gather_object`` to rank 0, which returns the complete list; every other rank returns
                             its own results and ``None`` for the rest -- the responsibilities travel once, to the rank
                             that writes them out (what ``vbx_amd.vbhmm`` and ``bench.py`` want)
      gather=False           nothing is exchanged: every rank returns its own results, ``None`` elsewhere

    Gathered results travel in rounds of at most ``gather_chunk_bytes`` (64 MB) per rank.

    The reference's 'auxiliary function has decreased' warning (VBx.py:123-124) is printed by the rank that ran the
    recording, once.

    ``run_shard(items, maxIters, epsilon)`` defaults to the HIP path; the CPU test-suite injects the oracle here."""
    import torch.distributed as dist
    if gather not in (True, False, 'root', 'all'):
        raise ValueError(f"gather={gather!r}: expected True / 'all' (every rank gets everything), 'root' (rank 0 does) or False")
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    # Who runs what follows from the shapes alone, so a rank normalises (copies, checks, converts) only the recordings it
    # owns: host work per rank is O(its shard), not O(corpus).  The one thing every rank must do for every recording is
    # the global-RNG draw of a ``gamma=None`` initialisation (VBx.py:79-83): the draws come from ONE stream in list order,
    # so a rank that skipped a foreign recording's draw would initialise its own later recordings differently from a
    # single process.  Recordings that bring their gamma cost a foreign rank nothing.
    shapes = [_shape_of(r, defaults) for r in recordings]
    assignment = shard_recordings([t * s for t, s, _ in shapes], world)
    mine, items = [], {}
    if run_shard is None:
        def run_shard(sub, mi, eps):
            return run_shard_hip(sub, mi, eps, precision=precision)
    # What only the owner can find out (values the library refuses, a device error) must not leave the other ranks waiting
    # in a collective: the local work runs under a guard, the ranks agree on the outcome in one tiny exchange, and a failure
    # anywhere is raised everywhere (the owner re-raises its own exception, the others name the rank and the message).
    failure, local = None, []
    try:
        for b, rec in enumerate(recordings):
            if assignment[b] == rank:
                items[b] = _normalise(rec, defaults)                       # (draws, if it has to, at its place in the order)
                mine.append(b)
            elif shapes[b][2] is not None:
                np.random.gamma(shapes[b][2], size=(shapes[b][0], shapes[b][1]))     # keep the global stream in step; discard
        local = run_shard([items[b] for b in mine], int(maxIters), epsilon) if mine else []
    except Exception as exc:                                               # noqa: BLE001 (re-raised below, on every rank)
        if world == 1:
            raise
        failure = exc
    if world > 1:
        status = [None] * world
        dist.all_gather_object(status, None if failure is None else f'{type(failure).__name__}: {failure}')
        if failure is not None:
            raise failure
        bad = [(r, m) for r, m in enumerate(status) if m is not None]
        if bad:
            raise RuntimeError('VBx_batch_distributed: ' + '; '.join(f'rank {r} failed ({m})' for r, m in bad))
    local = {b: res for b, res in zip(mine, local)}
    merged = dict(local)
    if gather and world > 1:
        # The results travel in rounds of at most ``gather_chunk_bytes`` per rank (the responsibilities are 8 T S bytes per
        # recording: a corpus in ONE pickled object would double every rank's footprint and serialise for seconds); every
        # rank takes part in the same number of rounds -- the largest any rank needs, agreed on first.
        rounds, cur, size = [], {}, 0
        for b_, res in local.items():
            nbytes = sum(getattr(v, 'nbytes', 64) for v in res.values())
            if cur and size + nbytes > gather_chunk_bytes:
                rounds.append(cur)
                cur, size = {}, 0
            cur[b_] = res
            size += nbytes
        rounds.append(cur)
        counts = [None] * world
        dist.all_gather_object(counts, len(rounds))
        for r in range(max(counts)):
            payload = rounds[r] if r < len(rounds) else {}
            if gather in (True, 'all'):
                parts = [None] * world
                dist.all_gather_object(parts, payload)
            else:
                parts = [None] * world if rank == 0 else None
                dist.gather_object(payload, parts, dst=0)
            for p in parts or []:
                merged.update(p)
    return [(_as_tuple(merged[b], return_model, warn=b in local) if b in merged else None) for b in range(len(recordings))]
