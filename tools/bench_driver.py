#!/usr/bin/env python
"""End-to-end rate of the batched diarization driver (files in -> RTTM files out), MI355X vs the same driver on the
host's CPU oracles.

    python tools/bench_driver.py [--recordings 16] [--xvectors 1025] [--cpu-recordings 1]

The archive is synthetic but x-vector-like: every recording is the reference's example recording (the 1025 real
x-vectors of tests/golden/driver_split3.npz) resampled in blocks of consecutive x-vectors (speaker turns stay
intact) with a little Gaussian noise, cut or repeated to ``--xvectors``; models = the example's transform and PLDA
from the same fixture.  One JSON line: x-vectors/s and recordings/s end to end, the stage split (read / AHC
initialisation / VB-HMM batch / RTTM), and the ``cpu_baseline`` = the same driver with the oracle stages
(``oracle/``: NumPy + SciPy, one thread) on a bounded sample of the same recordings.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def make_archive(tmp, n_rec, n_xvec, seed=0):
    from vbx_amd import kaldi_formats as kf
    g = np.load(os.path.join(REPO, 'tests', 'golden', 'driver_split3.npz'))
    rng = np.random.default_rng(seed)
    base, T0 = g['xvecs'], g['xvecs'].shape[0]
    paths = dict(ark=os.path.join(tmp, 'bench.ark'), seg=os.path.join(tmp, 'bench.seg'), plda=os.path.join(tmp, 'plda'),
                 transform=os.path.join(tmp, 'transform.npz'))
    items, segs = [], []
    for r in range(n_rec):
        idx = []
        while len(idx) < n_xvec:                              # blocks of 20..80 consecutive x-vectors
            lo = int(rng.integers(0, T0 - 80))
            idx.extend(range(lo, lo + int(rng.integers(20, 80))))
        idx = np.array(idx[:n_xvec])
        x = base[idx] + 0.02 * np.abs(base).mean() * rng.standard_normal((n_xvec, base.shape[1])).astype(np.float32)
        for k in range(n_xvec):
            name = f'rec{r:03d}_{k:05d}'
            items.append((name, x[k]))
            segs.append((name, f'rec{r:03d}', 0.24 * k, 0.24 * k + 1.44))
    kf.write_vec_flt_ark(paths['ark'], items)
    kf.write_segments(paths['seg'], segs)
    kf.write_plda(paths['plda'], g['plda_mean'], g['plda_trans'], g['plda_psi'])
    np.savez(paths['transform'], mean1=g['mean1'], mean2=g['mean2'], lda=g['lda'])
    return paths


def argv_for(paths, out):
    return ['--init', 'AHC+VB', '--out-rttm-dir', out, '--xvec-ark-file', paths['ark'], '--segments-file', paths['seg'],
            '--xvec-transform', paths['transform'], '--plda-file', paths['plda'], '--threshold', '-0.015',
            '--lda-dim', '128', '--Fa', '0.3', '--Fb', '17', '--loopP', '0.99']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--recordings', type=int, default=16)
    ap.add_argument('--xvectors', type=int, default=1025)
    ap.add_argument('--cpu-recordings', type=int, default=1, help='recordings of the CPU-oracle baseline run (0: skip)')
    ap.add_argument('--precision', default='fp64')
    a = ap.parse_args()
    from vbx_amd import vbhmm
    with tempfile.TemporaryDirectory() as tmp:
        paths = make_archive(tmp, a.recordings, a.xvectors)
        quiet = dict(log=lambda *_: None)
        args = vbhmm.build_parser().parse_args(argv_for(paths, os.path.join(tmp, 'warm')) + ['--precision', a.precision])
        vbhmm.diarize(args, **quiet)                          # warm-up: library load, allocator, first launches
        args = vbhmm.build_parser().parse_args(argv_for(paths, os.path.join(tmp, 'gpu')) + ['--precision', a.precision])
        t0 = time.perf_counter()
        state, timing = vbhmm.diarize(args, **quiet)
        wall = time.perf_counter() - t0
        line = {'metric': 'diarization driver, files in -> RTTM out (AHC+VB, x-vectors given)', 'unit': 'x-vectors/s',
                'value': a.recordings * a.xvectors / wall, 'recordings_per_s': a.recordings / wall, 'seconds': wall,
                'config': {'workload': f'{a.recordings} recordings x {a.xvectors} x-vectors (256-d), lda 128, Fa 0.3 Fb 17 loopP 0.99, '
                                       f'VB-HMM {a.precision}', 'speakers_init': [int(len(set(st['labels1st']))) for st in state.values()][:8]},
                'stages_s': {k: round(timing[k], 4) for k in ('read', 'project', 'ahc', 'vb', 'rttm')},
                'vb_iterations': [st['n_iters'] for st in state.values()][:8]}
        if a.cpu_recordings > 0:
            sys.path.insert(0, os.path.join(REPO, 'tests'))
            from test_driver import OracleStages            # the CPU checkers behind the driver's stage interface (one thread)

            small = make_archive(os.path.join(tmp), a.cpu_recordings, a.xvectors)          # same seed: the first recordings
            args = vbhmm.build_parser().parse_args(argv_for(small, os.path.join(tmp, 'cpu')))
            t0 = time.perf_counter()
            _, tc = vbhmm.diarize(args, stages=OracleStages(), **quiet)
            cw = time.perf_counter() - t0
            line['cpu_baseline'] = {'value': a.cpu_recordings * a.xvectors / cw, 'unit': 'x-vectors/s', 'cores': 1, 'kind': 'port',
                                    'sample': f'{a.cpu_recordings} of the same recordings through the same driver with the oracle '
                                              f'stages ({cw:.1f} s)', 'stages_s': {k: round(tc[k], 4) for k in ('read', 'project', 'ahc', 'vb', 'rttm')}}
            line['speedup_vs_cpu_baseline'] = line['value'] / line['cpu_baseline']['value']
        print(json.dumps(line))


if __name__ == '__main__':
    main()
