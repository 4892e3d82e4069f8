"""Is a recording's result independent of what else is in its batch -- for a library built one way or another?

Three recordings on one shared rho with the same hyper-parameters (T = 60 000: more workgroups than the chip holds at
once, so that workgroups in every phase share a CU) must agree bit for bit; the tiles whose partial sums differ are counted.
Run on A/B builds of the library (tools/build_variants.sh):
    cutbp  -DVBX_CUT_VIA_BPERMUTE    chunk_post's product at the cut with __shfl_xor, as round 3 had it: the compiler then
                                     vectorises the product into v_pk_fma_f32 ... op_sel:[0,1,0] and the build FAILS
    allbp  -DVBX_XOR_VIA_BPERMUTE    every add_xor / max_xor through __shfl_xor: fails the same way
and on the production library, which must not.  (Round 4 also ran builds that computed the reduction both ways in the
kernel and compared -- ds_bpermute_b32 and v_permlane*_swap agreed in all 50 652 reductions -- and builds with the
reduction as fixed machine code around the permutes: none of those failed, none held the packed form.)

usage: VBX_AMD_LIB=vbx_amd/csrc/libvbx_hip_<tag>.so python tools/hazard/bpermute_compare.py [T S n_rec precision reps]
"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from vbx_amd import _capi
from vbx_amd.synth import make_recording

T, S, n_rec = (int(a) for a in (sys.argv[1:4] + ['60000', '30', '3'][len(sys.argv[1:4]):]))
precision = sys.argv[4] if len(sys.argv) > 4 else 'fp32-split'
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 6
ctx = _capi.Context(0)
lib = ctx._lib
lib.vbx_debug_fetch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
lib.vbx_debug_fetch.restype = C.c_longlong
X, Phi, _ = make_recording(T, S, seed=3, kappa=0.05)
g0 = np.random.default_rng(4).gamma(1.0, size=(T, S)); g0 /= g0.sum(1, keepdims=True)
for rep in range(reps):
    b = _capi.Batch(ctx, [T] * n_rec, [S] * n_rec, 128, precision=precision, max_iters=2)
    if b.streams != 1: b.set_option(_capi.OPT_STREAMS, 1)
    for k in range(n_rec):
        if k: b.set_recording_shared(k, 0, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
        else: b.set_recording(k, X, Phi, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
    b.run(2, -np.inf)
    res = [b.result(k) for k in range(n_rec)]
    gemm = b.gemm
    nt = (T + 127) // 128
    Sp = 16
    while Sp < S: Sp *= 2
    ep = np.empty((n_rec, nt, Sp), dtype=np.float64)        # per-tile 'entered' partial sums of the last iteration
    assert lib.vbx_debug_fetch(b._h, 2, ep.ctypes.data_as(C.c_void_p), ep.nbytes) == ep.nbytes
    b.close()
    differ = sum(not np.array_equal(res[k]['gamma'], res[0]['gamma']) for k in range(1, n_rec))
    tiles = sum(int(np.count_nonzero(np.abs(ep[k] - ep[0]).max(1))) for k in range(1, n_rec))
    print(f'rep {rep} ({precision}, gemm {gemm}): recordings whose gamma differs from recording 0: {differ}, tiles whose partial sums differ: {tiles} of {(n_rec - 1) * nt}', flush=True)
