import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vbx_amd import _capi
from vbx_amd.synth import make_recording
ctx = _capi.Context(0)
T, S, n_rec, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
shared = len(sys.argv) > 5 and sys.argv[5] == 'shared'
X, Phi, _ = make_recording(T, S, seed=3, kappa=0.05)
g0 = np.random.default_rng(4).gamma(1.0, size=(T, S)); g0 /= g0.sum(1, keepdims=True)
b = _capi.Batch(ctx, [T] * n_rec, [S] * n_rec, 128, precision='fp32-split', max_iters=iters)
if b.streams != 1: b.set_option(_capi.OPT_STREAMS, 1)
for k in range(n_rec):
    if shared and k: b.set_recording_shared(k, 0, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
    else: b.set_recording(k, X, Phi, np.ones(S) / S, g0, 0.9, 0.3, 17.0)
b.run(iters, -np.inf)
rs = [b.result(k) for k in range(n_rec)]
b.close()
tag = f"mask={os.environ.get('VBX_AMD_SPLIT_MASK')} poison={os.environ.get('VBX_AMD_POISON')} T={T} S={S} n={n_rec} it={iters} {'shared' if shared else 'private'}"
print(tag, 'Li', [float(r['Li'][-1]) for r in rs], 'alpha dev vs rec0', [float(np.abs(r['alpha'] - rs[0]['alpha']).max()) for r in rs],
      'gamma dev', [float(np.abs(r['gamma'] - rs[0]['gamma']).max()) for r in rs], flush=True)
