#!/usr/bin/env python
"""What tools/profile_bench.sh puts under rocprofv3: the bench workload (64 recordings x T=10 000 x S=30 by default)
uploaded once, then ``--iters`` EM iterations in one vbx_batch_run -- no probe passes, no CPU baseline, so that every
launch in the trace belongs to the iteration loop (plus one upload pass and one gamma write-out)."""
import argparse
import os
import sys

os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--T', type=int, default=10000)
    ap.add_argument('--S', type=int, default=30)
    ap.add_argument('--D', type=int, default=128)
    ap.add_argument('--precision', default='fp32')
    ap.add_argument('--streams', type=int, default=None)
    ap.add_argument('--iters', type=int, default=12)
    ap.add_argument('--sweep', choices=['private', 'shared'], default=None,
                    help='the nine-point Fa/Fb sweep over ONE recording of --T x --S (bench.make_sweep_batch)')
    a = ap.parse_args()
    from vbx_amd import _capi
    ctx = _capi.Context(0)
    if a.sweep:
        from bench import make_sweep_batch, SWEEP_POINTS
        b = make_sweep_batch(ctx, a.T, a.S, a.D, a.precision, a.iters + 4, a.sweep == 'shared', streams=1)   # (kernel-level figures: one stream)
        a.batch = len(SWEEP_POINTS)
    else:
        b = make_batch(ctx, a.batch, a.T, a.S, a.D, a.precision, 0, a.iters + 4, streams=a.streams)
    b.run(a.iters, -np.inf)
    print('workload', f'batch={a.batch} T={a.T} S={a.S} D={a.D} precision={a.precision}' + (f' sweep={a.sweep}' if a.sweep else ''))
    print('iterations', a.iters, 'ms per iteration', b.last_run_ms()[0] / a.iters, 'streams', b.streams)
    b.close()


if __name__ == '__main__':
    main()
