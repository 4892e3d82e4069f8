"""fp32 vs fp64 (device) on the ill-conditioned toys of the S > 256 shape tests: which path, which T, which iteration."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vbx_amd
from vbx_amd.synth import make_recording
for S, T, D in ((256, 390, 128), (257, 390, 128), (257, 700, 128), (256, 700, 128), (200, 390, 128), (400, 390, 96), (512, 390, 64), (1000, 390, 40)):
    X, Phi, _ = make_recording(T, S, D=D, seed=S + D, kappa=0.1)
    g0 = np.random.default_rng(S * 7 + D).gamma(1.0, size=(T, S))
    g0 /= g0.sum(1, keepdims=True)
    for it in (1, 2, 3):
        kw = dict(loopProb=0.9, Fa=0.3, Fb=17.0, pi=S, gamma=g0, maxIters=it, epsilon=-1e300)
        g64, p64, L64 = vbx_amd.VBx(X, Phi, precision='fp64', **kw)
        g32, p32, L32 = vbx_amd.VBx(X, Phi, precision='fp32', **kw)
        os.environ['VBX_AMD_FB_ALGO'] = 'sequential'
        g32s, _, _ = vbx_amd.VBx(X, Phi, precision='fp32', **kw)
        del os.environ['VBX_AMD_FB_ALGO']
        print(f'S={S} T={T} it={it}: |g32-g64| {np.abs(g32 - g64).max():.2e}  seq-path {np.abs(g32s - g64).max():.2e}  pi {np.abs(p32 - p64).max():.2e}  maxgamma-min {g64.max(1).min():.3f}')
