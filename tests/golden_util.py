"""Helpers to rebuild the inputs of a golden synthetic case (tests/golden/synth_cases.npz)."""
import numpy as np

from vbx_amd.synth import make_recording


def case_inputs(case):
    """-> (X, Phi, kwargs) exactly as tests/golden/make_golden.py passed them to the reference."""
    T, S, seed, kappa = case['gen']
    X, Phi, _ = make_recording(int(T), int(S), seed=int(seed), kappa=float(kappa))
    chk = case['X_checksum']
    assert np.allclose([X.sum(), (X ** 2).sum(), Phi.sum()], chk, rtol=0, atol=1e-9 * max(1.0, abs(chk[1]))), \
        'synthetic generator drifted from the committed golden fixtures'
    kw = {}
    for key, val in case.items():
        if not key.startswith('kw_'):
            continue
        name = key[3:]
        if name in ('maxIters',):
            kw[name] = int(val)
        elif name == 'pi' and val.ndim == 0:
            kw[name] = int(val)
        elif val.ndim == 0:
            kw[name] = float(val)
        else:
            kw[name] = np.array(val)
    if 'gamma' not in kw:
        kw['gamma'] = None
    return X, Phi, kw
