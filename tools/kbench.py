#!/usr/bin/env python
"""Kernel-level timing of one library build on the bench workload (64 recordings x T=10 000 x S=30, f32 by default):
per-kernel HIP-event averages on ONE stream, then ms per iteration on the library's default streams.
usage: [VBX_AMD_LIB=...] tools/kbench.py [--batch 64] [--T 10000] [--S 30] [--precision fp32] [--iters 40]"""
import argparse
import json
import os
import sys
import time

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
    ap.add_argument('--iters', type=int, default=40)
    ap.add_argument('--sweep', choices=['private', 'shared'], default=None,
                    help='the nine-point Fa/Fb sweep over ONE recording (bench.make_sweep_batch) instead of --batch recordings')
    ap.add_argument('--tag', default=os.path.basename(os.environ.get('VBX_AMD_LIB', 'default')))
    args = ap.parse_args()
    from vbx_amd import _capi
    ctx = _capi.Context(0)
    n = args.iters
    out = {'tag': args.tag}
    if args.sweep:
        from bench import make_sweep_batch
        b = make_sweep_batch(ctx, args.T, args.S, args.D, args.precision, 3 * n + 8, args.sweep == 'shared', streams=1)
    else:
        b = make_batch(ctx, args.batch, args.T, args.S, args.D, args.precision, 0, 3 * n + 8, streams=1)
    b.run(4, -np.inf)
    b.profile_kernels(None)
    b.run(n, -np.inf)
    out['kernels_us'] = {k: round(1e3 * ms / c, 1) for k, (ms, c) in b.kernel_times().items() if c}
    b.profile_kernels([])
    b.run(n, -np.inf)
    out['one_stream_ms_per_iter'] = round(b.last_run_ms()[0] / n, 4)
    out['elbo_rec0'] = float(b.result(0, want_gamma=False, want_model=False)['Li'][-1])
    b.close()
    if args.sweep:
        # the same sweep on the streams the library's sweep API would pick (vbx_amd.batch.sweep_streams; VBX_AMD_SWEEP_STREAMS)
        b = make_sweep_batch(ctx, args.T, args.S, args.D, args.precision, 2 * n + 8, args.sweep == 'shared')
        b.run(4, -np.inf)
        t0 = time.perf_counter()
        b.run(n, -np.inf)
        out['default_streams'] = b.streams
        out['default_ms_per_iter'] = round(1e3 * (time.perf_counter() - t0) / n, 4)
        out['elbo_rec0_default'] = float(b.result(0, want_gamma=False, want_model=False)['Li'][-1])
        b.close()
        print(json.dumps(out))
        return
    b = make_batch(ctx, args.batch, args.T, args.S, args.D, args.precision, 0, 2 * n + 8)
    b.run(8, -np.inf)
    t0 = time.perf_counter()
    b.run(n, -np.inf)
    out['default_streams'] = b.streams
    out['default_ms_per_iter'] = round(1e3 * (time.perf_counter() - t0) / n, 4)
    b.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
