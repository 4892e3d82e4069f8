"""Multi-rank path on CPU: recordings shard over the ranks of a world_size-2 ``gloo`` group.

The data path has no collective: every rank runs its shard independently and one
``all_gather_object`` merges the small per-recording results.  Here the per-shard runner is the CPU
oracle (injected through ``run_shard``; the product default is the HIP path, which needs a GPU),
so what is tested is exactly the host logic bench.py / VBx_batch_distributed use on 8 GPUs:
the deterministic LPT assignment, identical RNG draws on every rank, the gather, the ordering.
"""
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _recordings():
    from vbx_amd.synth import make_recording
    recs = []
    for k, (T, S) in enumerate([(300, 5), (120, 3), (450, 6), (64, 2), (200, 4)]):
        X, Phi, _ = make_recording(T, S, D=32, seed=40 + k, kappa=0.05)
        recs.append(dict(X=X, Phi=Phi, pi=S, loopProb=0.9 if k % 2 else 0.8))
    return recs


def _oracle_shard(items, maxIters, epsilon):
    from oracle import vbx_oracle                               # checker standing in for the GPU
    out = []
    for it in items:
        g, pi, Li, alpha, invL = vbx_oracle.VBx(it['X'], it['Phi'], loopProb=it['loopProb'], Fa=it['Fa'],
                                                Fb=it['Fb'], pi=it['pi'], gamma=it['gamma'], maxIters=maxIters,
                                                epsilon=epsilon, return_model=True, alpha=it['alpha'],
                                                invL=it['invL'])
        out.append({'gamma': g, 'pi': pi, 'Li': np.array([row[0] for row in Li]), 'n_iters': len(Li),
                    'warned': False, 'alpha': alpha, 'invL': invL})
    return out


def assignment_of(recs):
    from vbx_amd.batch import shard_recordings
    return shard_recordings([r['X'].shape[0] * r['pi'] for r in recs], 2)


def _worker(rank, world, port, outdir):
    sys.path.insert(0, REPO)
    import torch.distributed as dist
    from vbx_amd.batch import VBx_batch_distributed
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        np.random.seed(7)                                       # gamma=None draws must agree on every rank
        res = VBx_batch_distributed(_recordings(), maxIters=4, epsilon=-np.inf, Fa=0.3, Fb=17.0,
                                    run_shard=_oracle_shard, return_model=True, gather='all')
        np.random.seed(7)
        tiny = VBx_batch_distributed(_recordings(), maxIters=4, epsilon=-np.inf, Fa=0.3, Fb=17.0, run_shard=_oracle_shard,
                                     return_model=True, gather='all', gather_chunk_bytes=1)   # one recording per round, ragged
        for x, y in zip(tiny, res):                              # round counts over the ranks (3 and 2): same results
            assert all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(x, y))
        np.random.seed(7)
        at_root = VBx_batch_distributed(_recordings(), maxIters=4, epsilon=-np.inf, Fa=0.3, Fb=17.0,
                                        run_shard=_oracle_shard, return_model=True, gather='root',
                                        gather_chunk_bytes=20000)   # gathers to rank 0, a few recordings per round
        np.random.seed(7)
        default = VBx_batch_distributed(_recordings(), maxIters=4, epsilon=-np.inf, Fa=0.3, Fb=17.0,
                                        run_shard=_oracle_shard, return_model=True)       # default (True): every rank, all
        assert all(r is not None for r in default)
        for x, y in zip(default, res):
            assert all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(x, y))
        np.random.seed(7)
        local_only = VBx_batch_distributed(_recordings(), maxIters=4, epsilon=-np.inf, Fa=0.3, Fb=17.0,
                                           run_shard=_oracle_shard, gather=False)
        mine = [b for b, r in enumerate(local_only) if r is not None]
        have = [b for b, r in enumerate(at_root) if r is not None]
        assert have == (list(range(len(res))) if rank == 0 else mine), (rank, have, mine)
        for b in have:
            assert all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(at_root[b], res[b]))
        # a rank normalises only what it owns; foreign recordings cost it the RNG draw (gamma=None) or nothing (gamma given)
        from vbx_amd import batch as vb
        mixed = _recordings()
        g_rng = np.random.default_rng(3)
        for k in (0, 3):
            T, S = mixed[k]['X'].shape[0], mixed[k]['pi']
            g = g_rng.gamma(1.0, size=(T, S))
            mixed[k]['gamma'] = g / g.sum(1, keepdims=True)
        calls, real = [], vb._normalise
        vb._normalise = lambda rec, defaults: (calls.append(rec['X'].shape[0]), real(rec, defaults))[1]
        try:
            np.random.seed(11)
            part = VBx_batch_distributed(mixed, maxIters=3, epsilon=-np.inf, Fa=0.3, Fb=17.0, run_shard=_oracle_shard, gather=False)
        finally:
            vb._normalise = real
        assert sorted(calls) == sorted(mixed[b]['X'].shape[0] for b in mine), (rank, calls, mine)
        np.savez(os.path.join(outdir, f'mixed{rank}.npz'), **{f'g{b}': r[0] for b, r in enumerate(part) if r is not None})
        # a recording that cannot run fails on EVERY rank, not only on its owner (the others used to wait in the gather):
        # (i) what the shapes show -- a gamma that does not fit -- before anything runs, the same AssertionError everywhere
        bad = _recordings()
        bad[2]['gamma'] = np.ones((bad[2]['X'].shape[0], bad[2]['pi'] + 1))
        try:
            VBx_batch_distributed(bad, maxIters=2, epsilon=-np.inf, Fa=0.3, Fb=17.0, run_shard=_oracle_shard, gather='all')
            raise SystemExit('a gamma of the wrong width was accepted')
        except AssertionError:
            pass
        # (ii) what only the owner finds out (here: its shard runner raises for one recording): the owner re-raises its own
        # exception, the other rank raises a RuntimeError naming it -- nobody enters the gather
        def failing_shard(items, mi, eps):
            if any(it['X'].shape[0] == 450 for it in items):
                raise ValueError('recording of 450 frames refused')
            return _oracle_shard(items, mi, eps)
        owner_450 = int(assignment_of(_recordings())[2])
        try:
            np.random.seed(7)
            VBx_batch_distributed(_recordings(), maxIters=2, epsilon=-np.inf, Fa=0.3, Fb=17.0, run_shard=failing_shard, gather='root')
            raise SystemExit('a failing shard went unnoticed')
        except ValueError:
            assert rank == owner_450
        except RuntimeError as exc:
            assert rank != owner_450 and f'rank {owner_450} failed' in str(exc) and '450 frames refused' in str(exc), str(exc)
        np.random.seed(7)                                       # and the group is still usable afterwards
        again = VBx_batch_distributed(_recordings(), maxIters=4, epsilon=-np.inf, Fa=0.3, Fb=17.0, run_shard=_oracle_shard,
                                      return_model=True, gather='all')
        for x, y in zip(again, res):
            assert all(np.array_equal(np.asarray(u), np.asarray(v)) for u, v in zip(x, y))
        np.savez(os.path.join(outdir, f'rank{rank}.npz'), mine=np.array(mine),
                 **{f'g{b}': r[0] for b, r in enumerate(res)}, **{f'pi{b}': r[1] for b, r in enumerate(res)},
                 **{f'L{b}': np.array(r[2]) for b, r in enumerate(res)},
                 **{f'a{b}': r[3] for b, r in enumerate(res)})
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_matches_single_process(tmp_path):
    import torch.multiprocessing as mp
    from vbx_amd.batch import VBx_batch_distributed, shard_recordings
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    np.random.seed(7)
    single = VBx_batch_distributed(_recordings(), maxIters=4, epsilon=-np.inf, Fa=0.3, Fb=17.0,
                                   run_shard=_oracle_shard, return_model=True)     # no group: world = 1
    ranks = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
    recs = _recordings()
    assign = shard_recordings([r['X'].shape[0] * r['pi'] for r in recs], world)
    for r in range(world):
        assert sorted(ranks[r]['mine'].tolist()) == [b for b in range(len(recs)) if assign[b] == r]
        for b, ref in enumerate(single):
            assert np.array_equal(ranks[r][f'g{b}'], ref[0])
            assert np.array_equal(ranks[r][f'pi{b}'], ref[1])
            assert np.array_equal(ranks[r][f'L{b}'], np.array(ref[2]))
            assert np.array_equal(ranks[r][f'a{b}'], ref[3])
    assert set(ranks[0]['mine'].tolist()) | set(ranks[1]['mine'].tolist()) == set(range(len(recs)))
    assert not set(ranks[0]['mine'].tolist()) & set(ranks[1]['mine'].tolist())
    # the mixed list (two recordings bring their gamma, three draw it): every rank's own results are what one process gets
    mixed = _recordings()
    g_rng = np.random.default_rng(3)
    for k in (0, 3):
        g = g_rng.gamma(1.0, size=(mixed[k]['X'].shape[0], mixed[k]['pi']))
        mixed[k]['gamma'] = g / g.sum(1, keepdims=True)
    np.random.seed(11)
    one = VBx_batch_distributed(mixed, maxIters=3, epsilon=-np.inf, Fa=0.3, Fb=17.0, run_shard=_oracle_shard)
    seen = set()
    for r in range(world):
        with np.load(tmp_path / f'mixed{r}.npz') as z:
            for key in z.files:
                assert np.array_equal(z[key], one[int(key[1:])][0]), (r, key)
                seen.add(int(key[1:]))
    assert seen == set(range(len(mixed)))


def test_lpt_assignment_is_balanced_and_deterministic():
    from vbx_amd.batch import shard_recordings
    costs = [10_000 * 30] * 64
    a = shard_recordings(costs, 8)
    assert np.bincount(a, minlength=8).tolist() == [8] * 8              # BASELINE config 4: 64 recordings / 8 GPUs
    ragged = np.random.default_rng(0).integers(1_000, 200_000, 37) * 30
    a1, a2 = shard_recordings(ragged, 4), shard_recordings(ragged, 4)
    assert np.array_equal(a1, a2)
    load = np.bincount(a1, weights=ragged, minlength=4)
    assert load.max() <= ragged.sum() / 4 + ragged.max()                # LPT bound
    assert shard_recordings([5, 1], 4).tolist() == [0, 1]


def test_bench_rejects_mismatched_world_size():
    import subprocess
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    res = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '4'], env=env,
                         capture_output=True, text=True)
    assert res.returncode != 0 and 'WORLD_SIZE' in (res.stderr + res.stdout)


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus N` without a torchrun environment (how the driver calls it) becomes N ranks."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--dry-run'], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    ranks = [json.loads(line) for line in res.stdout.splitlines() if line.startswith('{')]
    assert sorted(r['rank'] for r in ranks) == [0, 1] and all(r['world'] == 2 for r in ranks), res.stdout
