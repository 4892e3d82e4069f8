"""Instrumentation: per-workgroup phase timeline of chunk_post (library built with -DVBX_PHASE_CLOCKS).

    tools/build_variants.sh clk:"-DVBX_PHASE_CLOCKS"
    VBX_AMD_LIB=vbx_amd/csrc/libvbx_hip_clk.so python tools/phase_timeline.py
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbx_amd import _capi  # noqa: E402
from vbx_amd.synth import make_recording  # noqa: E402


def main(nrec=64, T=10000, S=30, precision='fp32', out='gpurun_out/phase_timeline.npy'):
    ctx = _capi.Context(0)
    lib = _capi.load()
    batch = _capi.Batch(ctx, [T] * nrec, [S] * nrec, 128, precision=precision, max_iters=6)
    batch.set_option(_capi.OPT_STREAMS, 1)           # tile numbers index the stamp table: one stream group
    X, Phi, _ = make_recording(T, S, seed=1, kappa=0.05)
    g0 = np.random.default_rng(2).gamma(1.0, size=(T, S))
    g0 /= g0.sum(1, keepdims=True)
    for j in range(nrec):
        batch.set_recording(j, X, Phi, np.ones(S) / S, g0, 0.99, 0.3, 17.0)
    batch.run(6, -np.inf)
    ntile = nrec * ((T + 127) // 128)
    buf = np.zeros((8192, 2, 8), np.int64)
    rc = lib.vbx_debug_clocks(buf.ctypes.data_as(C.c_void_p), buf.size)
    assert rc == 0, rc
    ntile = min(ntile, 8192)
    buf = buf[:ntile]
    np.save(out, buf)
    w0 = buf[:, 0, :]
    w1 = buf[:, 1, :]
    t0 = w0[:, 0].min()
    names = ['stage', 'half1', 'half2', 'wait', 'post', 'mfma']
    d = np.diff(w0[:, :7], axis=1)
    print('kernel span (cycles):', int(w0[:, 6].max() - t0))
    order = np.argsort(w0[:, 0])
    for q in range(0, ntile, max(1, ntile // 40)):
        k = order[q]
        print(f'blk {k:5d} hw {w0[k, 7]:08x} start {w0[k, 0] - t0:8d} ' +
              ' '.join(f'{n} {v:6d}' for n, v in zip(names, d[k])) + f'  life {w0[k, 6] - w0[k, 0]:7d}')
    print('median phases:', dict(zip(names, np.median(d, axis=0).astype(int))), 'life', int(np.median(w0[:, 6] - w0[:, 0])))
    print('second recorded wave :', dict(zip(names, np.median(np.diff(w1[:, :7], axis=1), axis=0).astype(int))))
    # HW_ID (gfx9 layout): wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]
    for nm, w in (('first recorded wave', w0), ('second recorded wave', w1)):
        hw = w[:, 7]
        print(nm, 'on SIMD 0..3:', np.bincount((hw >> 4) & 3, minlength=4).tolist(),
              ' wave slot histogram:', np.bincount(hw & 15, minlength=16).tolist())
    early = order[:1024]
    late = order[1024:]
    print('first 1024 started :', dict(zip(names, np.median(d[early], axis=0).astype(int))))
    print('the rest           :', dict(zip(names, np.median(d[late], axis=0).astype(int))))


if __name__ == '__main__':
    a = [int(v) for v in sys.argv[1:4]]          # [recordings T S [precision]]
    main(*a, *sys.argv[4:5]) if a else main()
