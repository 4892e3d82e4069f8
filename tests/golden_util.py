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


def load_config(name):
    """tests/golden/config_<name>.npz (tests/golden/make_golden_configs.py: BASELINE.json's configs at full size, outputs
    of the unmodified reference VBx()) as a dict keyed by the part of the key behind 'name/'."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', f'config_{name}.npz')
    return dict(np.load(path))


def config_inputs(cfg, prefix, g0_seed=None):
    """Regenerate (X, Phi, gamma0 or None) of a full-size fixture from the generator arguments stored under `prefix`;
    checksums guard against generator drift."""
    T, S, seed, kappa = cfg[prefix + '/gen']
    T, S = int(T), int(S)
    X, Phi, _ = make_recording(T, S, seed=int(seed), kappa=float(kappa))
    chk = cfg[prefix + '/X_checksum']
    assert np.allclose([X.sum(), (X ** 2).sum(), Phi.sum()], chk, rtol=0, atol=1e-9 * max(1.0, abs(chk[1]))), \
        'synthetic generator drifted from the committed golden fixtures'
    g0 = None
    if g0_seed is not None:
        g0 = np.random.default_rng(g0_seed).gamma(1.0, size=(T, S))
        g0 /= g0.sum(1, keepdims=True)
        gchk = cfg[prefix + '/g0_checksum']
        assert np.allclose([g0.sum(), (g0 ** 2).sum(), g0[T // 2].max()], gchk, rtol=1e-12, atol=0), \
            'initial responsibilities drifted from the committed golden fixtures'
    return X, Phi, g0


def config_diffs(cfg, tag, gamma, pi, Li, alpha=None, invL=None):
    """Largest deviations of a result from the reference outputs stored under `tag` (gamma on the sampled rows)."""
    rows = cfg[tag.split('/')[0] + '/rows']
    ref_Li = cfg[tag + '/Li']
    out = {
        'n_iters': (len(Li), len(ref_Li)),
        'gamma': float(np.abs(gamma[rows] - cfg[tag + '/gamma_rows']).max()),
        'gamma_colsum_rel': float((np.abs(gamma.sum(0) - cfg[tag + '/gamma_colsum']) /
                                   np.maximum(1.0, cfg[tag + '/gamma_colsum'])).max()),
        'pi': float(np.abs(pi - cfg[tag + '/pi']).max()),
    }
    n = min(len(Li), len(ref_Li))
    out['Li_rel'] = float(np.max(np.abs(np.asarray(Li[:n]) - ref_Li[:n]) / np.abs(ref_Li[:n]))) if n else 0.0
    if alpha is not None:
        ra = cfg[tag + '/alpha']
        out['alpha'] = float(np.abs(alpha - ra).max() / max(1.0, np.abs(ra).max()))
        out['invL_rel'] = float((np.abs(invL - cfg[tag + '/invL']) / cfg[tag + '/invL']).max())
    return out
