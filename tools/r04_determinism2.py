import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vbx_amd import _capi
from vbx_amd.synth import make_recording
ctx = _capi.Context(0)
T, S = int(sys.argv[1]), int(sys.argv[2])
X, Phi, _ = make_recording(T, S, seed=3, kappa=0.05)
g0 = np.random.default_rng(4).gamma(1.0, size=(T, S)); g0 /= g0.sum(1, keepdims=True)
def run(iters):
    b = _capi.Batch(ctx, [T], [S], 128, precision='fp32-split', max_iters=iters)
    b.set_recording(0, X, Phi, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
    b.run(iters, -np.inf)
    r = b.result(0); b.close(); return r
rs = [run(2) for _ in range(4)]
for r in rs[1:]:
    print(os.environ.get('VBX_AMD_SPLIT_MASK'), T, S, 'gamma', np.abs(r['gamma'] - rs[0]['gamma']).max(), 'alpha', np.abs(r['alpha'] - rs[0]['alpha']).max(),
          'invL', np.abs(r['invL'] - rs[0]['invL']).max(), 'rows differing in alpha', np.nonzero(np.abs(r['alpha'] - rs[0]['alpha']).max(1))[0][:10],
          'cols', np.nonzero(np.abs(r['alpha'] - rs[0]['alpha']).max(0))[0][:12])
