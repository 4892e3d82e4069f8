"""chunk_post's input checksums per tile (debug build -DVBX_DEBUG_INPUTS) next to its outputs: shared rho, split GEMM."""
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
for rep in range(4):
    b = _capi.Batch(ctx, [T] * n_rec, [S] * n_rec, 128, precision='fp32-split', max_iters=1)
    if b.streams != 1: b.set_option(_capi.OPT_STREAMS, 1)
    for k in range(n_rec):
        if k: b.set_recording_shared(k, 0, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
        else: b.set_recording(k, X, Phi, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
    b.run(1, -np.inf)
    def fetch(which, shape, dt):
        a = np.empty(shape, dtype=dt)
        got = lib.vbx_debug_fetch(b._h, which, a.ctypes.data_as(C.c_void_p), a.nbytes)
        assert got == a.nbytes, (which, got, a.nbytes)
        return a
    ep = fetch(2, (n_rec, nt, Sp), np.float64)
    dbg = fetch(5, (n_rec, nt, 8), np.float64)
    b.close()
    names = ['gbound(w1)', 'a pre-crossing', 'x pre-crossing SECOND half', 'x pre-crossing FIRST half', 'x at the cut (mv_w[1])', 'x post-crossing', 'sfl', 'qfl']
    for k in range(1, n_rec):
        bad = np.nonzero(np.abs(ep[k] - ep[0]).max(1))[0]
        for t in bad:
            diff = [names[j] for j in range(8) if dbg[k, t, j] != dbg[0, t, j]]
            print(f'rep {rep} rec {k} tile {t}: epart differs; inputs that differ: {diff}', flush=True)
        clean_in = np.nonzero(np.abs(dbg[k] - dbg[0]).max(1))[0]
        extra = [t for t in clean_in if t not in bad]
        print(f'rep {rep} rec {k}: {len(bad)} bad tiles; tiles whose inputs differ but outputs do not: {extra[:10]}', flush=True)
