"""The multi-rank PRODUCT path on the GPU (-m gpu): several processes, each with its own ``vbx_ctx`` (stream, device
arena), sharing the ONE device of a test box over a ``gloo`` rendezvous -- RCCL refuses two ranks on one GPU, and the
data path has no collective anyway (on a node every rank has its own GPU and the same code runs
over RCCL).  tests/test_distributed_gloo.py covers the host logic on CPU with the oracle injected; here the shards run
on the HIP kernels:

  * ``VBx_batch_distributed`` on two ranks == ``VBx_batch`` in one process, recording by recording;
  * ``python -m torch.distributed.run --nproc-per-node 2 -m vbx_amd.vbhmm ...`` writes the RTTM files the unchanged
    reference driver wrote for the same archive (tests/golden/driver_split3.npz), each exactly once.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _recordings():
    from vbx_amd.synth import make_recording
    recs = []
    for k, (T, S) in enumerate([(3000, 12), (700, 5), (5200, 30), (129, 3), (1500, 9), (2600, 40)]):
        X, Phi, _ = make_recording(T, S, seed=60 + k, kappa=0.05)
        recs.append(dict(X=X, Phi=Phi, pi=S, loopProb=0.99 if k % 2 else 0.9))
    return recs


def _worker(rank, world, port, outdir):
    sys.path.insert(0, REPO)
    os.environ['VBX_AMD_DEVICE'] = '0'                          # every rank on the one device
    import torch.distributed as dist
    from vbx_amd import _capi
    from vbx_amd.batch import VBx_batch_distributed
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        np.random.seed(11)                                      # gamma=None: the same draws on every rank
        res = VBx_batch_distributed(_recordings(), maxIters=6, epsilon=1e-6, Fa=0.3, Fb=17.0, return_model=True,
                                    gather='root')
        have = [b for b, r in enumerate(res) if r is not None]
        np.savez(os.path.join(outdir, f'rank{rank}.npz'), have=np.array(have), lib=np.array(_capi.library_path()),
                 loaded=np.array(_capi._lib is not None),
                 **{f'g{b}': res[b][0] for b in have}, **{f'pi{b}': res[b][1] for b in have},
                 **{f'L{b}': np.array(res[b][2]) for b in have}, **{f'a{b}': res[b][3] for b in have})
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_process(tmp_path):
    import torch.multiprocessing as mp
    from vbx_amd.batch import VBx_batch, shard_recordings
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    np.random.seed(11)
    single = VBx_batch(_recordings(), maxIters=6, epsilon=1e-6, Fa=0.3, Fb=17.0, return_model=True)
    recs = _recordings()
    assign = shard_recordings([r['X'].shape[0] * r['pi'] for r in recs], world)
    ranks = [np.load(tmp_path / f'rank{r}.npz') for r in range(world)]
    assert ranks[0]['have'].tolist() == list(range(len(recs)))                  # the gather ends at rank 0
    assert ranks[1]['have'].tolist() == [b for b in range(len(recs)) if assign[b] == 1]
    assert 0 < len(ranks[1]['have']) < len(recs)
    for r in range(world):
        assert bool(ranks[r]['loaded']) and str(ranks[r]['lib']).endswith('libvbx_hip.so')
        for b in ranks[r]['have'].tolist():
            # (a recording's arithmetic does not depend on which other recordings share its batch: fp64, same kernels)
            np.testing.assert_allclose(ranks[r][f'g{b}'], single[b][0], rtol=0, atol=1e-12, err_msg=f'rank {r} recording {b}')
            np.testing.assert_allclose(ranks[r][f'pi{b}'], single[b][1], rtol=0, atol=1e-13)
            assert len(ranks[r][f'L{b}']) == len(single[b][2])
            np.testing.assert_allclose(ranks[r][f'L{b}'].ravel(), np.array(single[b][2]).ravel(), rtol=1e-13)
            np.testing.assert_allclose(ranks[r][f'a{b}'], single[b][3], rtol=0, atol=1e-12)


def test_driver_under_torchrun_on_one_gpu(tmp_path):
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import test_driver as td
    paths = td._write_inputs(tmp_path)
    env = dict(os.environ, VBX_AMD_DEVICE='0', VBX_AMD_DIST_BACKEND='gloo', PYTHONFAULTHANDLER='1', PYTHONPATH=REPO + os.pathsep + os.environ.get('PYTHONPATH', ''))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), '-m', 'vbx_amd.vbhmm'] + td._argv(paths, ['--timing'])
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=REPO)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-3000:])
    timings = [json.loads(line) for line in res.stdout.splitlines() if line.startswith('{"read"') or line.startswith('{')]
    timings = [t for t in timings if 'world' in t]
    assert sorted(t['rank'] for t in timings) == [0, 1] and all(t['world'] == 2 for t in timings), res.stdout
    assert sorted(t['recordings'] for t in timings) == [1, 2]                  # recA on one rank, recB + recC on the other
    td._check_rttm(paths)


# ---- RCCL: only on a box with more than one GPU (the single-GPU test boxes skip; a multi-GPU lease tests RCCL, not gloo) ----
def _gpu_count():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _rccl_worker(rank, world, port, outdir):
    sys.path.insert(0, REPO)
    os.environ['VBX_AMD_DEVICE'] = str(rank)                    # one GPU per rank
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    from vbx_amd.batch import VBx_batch_distributed
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world,
                            device_id=torch.device('cuda', rank))
    try:
        one = torch.ones(1, device='cuda')
        dist.all_reduce(one)                                    # RCCL carries a byte
        assert int(one.item()) == world
        np.random.seed(11)
        res = VBx_batch_distributed(_recordings(), maxIters=6, epsilon=1e-6, Fa=0.3, Fb=17.0, return_model=True, gather='root')
        have = [b for b, r in enumerate(res) if r is not None]
        np.savez(os.path.join(outdir, f'rccl{rank}.npz'), have=np.array(have),
                 **{f'g{b}': res[b][0] for b in have}, **{f'pi{b}': res[b][1] for b in have})
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(_gpu_count() < 2, reason='RCCL needs one GPU per rank: fewer than two GPUs on this box')
def test_rccl_gather_to_root_on_a_multi_gpu_box(tmp_path):
    """Two ranks, two GPUs, backend nccl (= RCCL): an all_reduce, then VBx_batch_distributed(gather='root') == VBx_batch."""
    import torch.multiprocessing as mp
    from vbx_amd.batch import VBx_batch
    world, port = 2, _free_port()
    mp.spawn(_rccl_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    np.random.seed(11)
    single = VBx_batch(_recordings(), maxIters=6, epsilon=1e-6, Fa=0.3, Fb=17.0, return_model=True)
    root = np.load(tmp_path / 'rccl0.npz')
    assert root['have'].tolist() == list(range(len(single)))
    for b in range(len(single)):
        np.testing.assert_allclose(root[f'g{b}'], single[b][0], rtol=0, atol=1e-12)
        np.testing.assert_allclose(root[f'pi{b}'], single[b][1], rtol=0, atol=1e-13)


@pytest.mark.skipif(_gpu_count() < 2, reason='fewer than two GPUs on this box')
def test_bench_two_gpus_prints_a_compact_line_with_both_scaling_forms():
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    res = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
                          '--min-seconds', '0.1', '--batch', '8', '--full-out', ''], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('{')][-1]
    rec = json.loads(line)
    assert len(line) < 4096 and rec['n_gpus'] == 2 and rec['scaling'] == 'weak' and rec['value'] > 0
    assert rec['configs']['strong_scaling_form']['value'] > 0


def test_bench_n_rank_code_path_on_one_gpu():
    """bench.py's N > 1 path -- torchrun launch, barrier + MAX-over-ranks timing, the strong-scaling form next to the weak
    headline, rank 0 printing ONE compact line -- exercised on the single GPU of a test box through the test hook
    (backend gloo, both ranks on device 0).  Not a scaling measurement; the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(VBX_AMD_DIST_BACKEND='gloo', VBX_AMD_DEVICE='0')
    res = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2',
                          '--min-seconds', '0.05', '--batch', '6', '--full-out', ''], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, res.stdout                       # rank 0 alone speaks
    rec = json.loads(lines[0])
    assert len(lines[0]) < 4096 and rec['n_gpus'] == 2 and rec['scaling'] == 'weak' and rec['steps'] == 5 and rec['warmup'] == 2
    assert rec['value'] > 0 and abs(rec['value'] - 2 * 6 * 1e3 / rec['ms_per_step']) < 1e-3 * rec['value']
    strong = rec['configs']['strong_scaling_form']
    assert strong['value'] > 0 and abs(strong['value'] - 64 * 1e3 / strong['ms']) < 1e-3 * strong['value']
    assert 'TEST HOOK' in rec['config']['parallelism'] and rec['roofline']['frac'] > 0
    assert 'cpu_baseline' not in rec                         # (rank 0 at N = 1 only)
