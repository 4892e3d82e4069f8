"""Per-tile partial sums of chunk_post after ONE iteration: shared vs private rho, split GEMM (debugging aid)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vbx_amd import _capi
from vbx_amd.synth import make_recording
ctx = _capi.Context(0)
lib = ctx._lib
lib.vbx_debug_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
lib.vbx_debug_fetch.restype = C.c_longlong
T, S, n_rec = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
Sp = 16
while Sp < S: Sp *= 2
X, Phi, _ = make_recording(T, S, seed=3, kappa=0.05)
g0 = np.random.default_rng(4).gamma(1.0, size=(T, S)); g0 /= g0.sum(1, keepdims=True)
nt = (T + 127) // 128


def parts(shared):
    b = _capi.Batch(ctx, [T] * n_rec, [S] * n_rec, 128, precision='fp32-split', max_iters=1)
    if b.streams != 1: b.set_option(_capi.OPT_STREAMS, 1)
    for k in range(n_rec):
        if shared and k: b.set_recording_shared(k, 0, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
        else: b.set_recording(k, X, Phi, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
    b.run(1, -np.inf)
    out = {}
    for which, name, shape, dt in ((0, 'mpart', (n_rec * nt, Sp, 128), np.float32), (1, 'npart', (n_rec * nt, Sp), np.float32),
                                   (2, 'epart', (n_rec * nt, Sp), np.float64), (3, 'tllpart', (n_rec * nt,), np.float64)):
        a = np.empty(shape, dtype=dt)
        got = lib.vbx_debug_fetch(b._h, which, a.ctypes.data_as(C.c_void_p), a.nbytes)
        assert got == a.nbytes, (name, got, a.nbytes)
        out[name] = a.reshape((n_rec, nt) + shape[1:])
    b.close()
    return out


pr, sh = parts(False), parts(True)
for name in pr:
    for k in range(n_rec):
        d_ps = np.abs(pr[name][k] - sh[name][k]).reshape(nt, -1).max(1)
        d_p0 = np.abs(pr[name][k] - pr[name][0]).reshape(nt, -1).max(1)
        d_s0 = np.abs(sh[name][k] - sh[name][0]).reshape(nt, -1).max(1)
        bad = np.nonzero(d_s0)[0]
        print(f'{name:8s} rec {k}: private-vs-shared {d_ps.max():.3e} ({(d_ps > 0).sum()} tiles)  private rec vs rec0 {d_p0.max():.3e}  shared rec vs rec0 {d_s0.max():.3e} '
              f'({len(bad)} tiles: {bad[:12]})', flush=True)
        if name == 'mpart' and len(bad):
            t = bad[0]
            dd = np.abs(sh[name][k][t] - sh[name][0][t])
            print('   first bad tile', t, 'speakers', np.nonzero(dd.max(1))[0][:16], 'dims', np.nonzero(dd.max(0))[0][:40], 'values', sh[name][k][t][dd > 0][:6], sh[name][0][t][dd > 0][:6])
tot = sum(int((np.abs(sh['mpart'][k] - sh['mpart'][0]).reshape(nt, -1).max(1) > 0).sum()) for k in range(n_rec))
tot_p = sum(int((np.abs(pr['mpart'][k] - sh['mpart'][0]).reshape(nt, -1).max(1) > 0).sum()) for k in range(n_rec))
print('SUMMARY lib', os.environ.get('VBX_AMD_LIB', 'default'), 'mask', os.environ.get('VBX_AMD_SPLIT_MASK'), 'bad tiles among sharers', tot, 'private tiles differing from shared rec0', tot_p, flush=True)
print('NANS', {n: (int(np.isnan(pr[n]).sum()), int(np.isnan(sh[n]).sum())) for n in pr}, flush=True)
