import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbx_amd.batch import VBx_batch
from vbx_amd.synth import make_recording
n, T, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
recs = []
for b in range(n):
    X, Phi, _ = make_recording(T, S, seed=b, kappa=0.05, dtype=np.float32)
    g = np.random.default_rng(10_000 + b).gamma(1.0, size=(T, S)).astype(np.float32)
    g /= g.sum(1, keepdims=True)
    recs.append(dict(X=X, Phi=Phi, pi=S, gamma=g, loopProb=0.99, Fa=0.3, Fb=17.0))
ts = []
for _ in range(7):
    t0 = time.perf_counter(); out = VBx_batch(recs, maxIters=iters, epsilon=-np.inf); ts.append(time.perf_counter() - t0)
print(f'VBx_batch n={n} T={T} S={S} iters={iters} VBX_AMD_STREAMS={os.environ.get("VBX_AMD_STREAMS")}: call {1e3*min(ts[1:]):.3f} ms (median {1e3*sorted(ts[1:])[3]:.3f}); checksum {sum(float(o[0].sum()) for o in out):.6f}')
