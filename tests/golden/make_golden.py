#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the authoring container, where /root/reference exists; the GPU box
never sees the reference and only reads the committed .npz / .json files.

What it produces
  es2005a.npz      inputs to VBx() for the reference's only end-to-end example
                   (exp/ES2005a.ark + models/ResNet101_16kHz, flags of run_example.sh:23-34)
                   captured at the call site vbhmm.py:154-158 while running the
                   UNMODIFIED reference vbhmm.py, plus the reference VBx() outputs
                   for (maxIters=40, eps=1e-6) and (maxIters=10, eps=1e-4), the x-vector
                   timing table and the RTTM segments the reference wrote.
  synth_cases.npz  small synthetic recordings (vbx_amd.synth) and the reference VBx()
                   outputs on them (inputs are stored as generator arguments + checksum): soft/easy data, gamma=None (global RNG), pi vector,
                   warm start (alpha/invL), S=1, T=1, T=2, loopProb in {0,1}, maxIters=0.
  fb_cases.npz     forward_backward() known answers on random log-likelihoods.

Three import shims are needed because kaldi_io, h5py and fastcluster are not installed
here; they are created in a temp dir and never committed as product code.
"""
import io
import os
import runpy
import sys
import tempfile
import textwrap
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)

from vbx_amd.synth import make_recording, make_lls  # noqa: E402

SHIMS = {
    'kaldi_io/__init__.py': '''
        import struct, numpy as np
        class BadSampleSize(Exception): pass
        class UnknownMatrixHeader(Exception): pass
        def open_or_fd(f, mode='rb'):
            return open(f, mode) if isinstance(f, str) else f
        def read_vec_flt_ark(path):
            with open(path, 'rb') as fd:
                while True:
                    key = b''
                    while True:
                        ch = fd.read(1)
                        if ch == b'' or ch == b' ':
                            break
                        key += ch
                    if not key:
                        return
                    assert fd.read(2) == b'\\x00B'
                    kind = fd.read(3)
                    size = 4 if kind == b'FV ' else 8
                    assert fd.read(1) == b'\\x04'
                    n = struct.unpack('<i', fd.read(4))[0]
                    vec = np.frombuffer(fd.read(n * size), dtype='float32' if size == 4 else 'float64')
                    yield key.decode(), vec
    ''',
    'kaldi_io/kaldi_io.py': '''
        def _read_compressed_mat(*a, **k): raise NotImplementedError
        def _read_mat_ascii(*a, **k): raise NotImplementedError
    ''',
    'h5py.py': '''
        import numpy as np
        # transform.h5 holds three contiguous float64 datasets (SURVEY.md §2, measured offsets)
        _LAYOUT = {'mean1': (2048, (256,)), 'mean2': (4096, (128,)), 'lda': (5120, (256, 128))}
        class File:
            def __init__(self, path, mode='r'):
                self._raw = open(path, 'rb').read()
            def __enter__(self): return self
            def __exit__(self, *a): return False
            def __getitem__(self, name):
                off, shape = _LAYOUT[name]
                n = int(np.prod(shape))
                return np.frombuffer(self._raw, dtype='<f8', count=n, offset=off).reshape(shape)
    ''',
    'fastcluster.py': '''
        from scipy.cluster.hierarchy import linkage as _linkage
        def linkage(y, method='single', preserve_input=True):
            return _linkage(y, method=method)
    ''',
    # "VBx" module that vbhmm.py:45 imports: records the call and forwards to the reference.
    'VBx.py': '''
        import importlib.util, numpy as np
        _spec = importlib.util.spec_from_file_location('_ref_VBx', '%(ref)s/VBx/VBx.py')
        _ref = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_ref)
        CALLS = []
        def VBx(X, Phi, **kw):
            out = _ref.VBx(X, Phi, **kw)
            CALLS.append((np.array(X), np.array(Phi), dict(kw), out))
            return out
    ''' % {'ref': REF},
}


def ref_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('_ref_VBx_direct', f'{REF}/VBx/VBx.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def run_reference_vbhmm(tmp):
    for rel, src in SHIMS.items():
        path = os.path.join(tmp, 'shims', rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            f.write(textwrap.dedent(src))
    out_dir = os.path.join(tmp, 'rttm')
    argv = ['vbhmm.py', '--init', 'AHC+VB', '--out-rttm-dir', out_dir,
            '--xvec-ark-file', f'{REF}/exp/ES2005a.ark',
            '--segments-file', f'{REF}/exp/ES2005a.seg',
            '--xvec-transform', f'{REF}/VBx/models/ResNet101_16kHz/transform.h5',
            '--plda-file', f'{REF}/VBx/models/ResNet101_16kHz/plda',
            '--threshold', '-0.015', '--lda-dim', '128',
            '--Fa', '0.3', '--Fb', '17', '--loopP', '0.99']      # run_example.sh:23-34
    old_argv, old_path = sys.argv, list(sys.path)
    sys.argv = argv
    sys.path[:0] = [os.path.join(tmp, 'shims'), f'{REF}/VBx']
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            runpy.run_path(f'{REF}/VBx/vbhmm.py', run_name='__main__')
        calls = sys.modules['VBx'].CALLS
    finally:
        sys.argv, sys.path[:] = old_argv, old_path
        for name in ('VBx', 'kaldi_io', 'kaldi_io.kaldi_io', 'h5py', 'fastcluster',
                     'diarization_lib', 'kaldi_utils'):
            sys.modules.pop(name, None)
    rttm = open(os.path.join(out_dir, 'ES2005a.rttm')).read()
    return calls, rttm


def parse_rttm(text):
    rows = []
    for line in text.strip().splitlines():
        f = line.split()
        rows.append((float(f[3]), float(f[4]), int(f[7])))
    return np.array(rows)


def li_array(Li):
    return np.array([row[0] for row in Li], dtype=np.float64)


def main():
    ref = ref_module()
    # ---------------------------------------------------------------- ES2005a
    with tempfile.TemporaryDirectory() as tmp:
        calls, rttm_text = run_reference_vbhmm(tmp)
    assert len(calls) == 1
    fea, Phi, kw, out40 = calls[0]
    qinit = np.array(kw['gamma'])
    g40, pi40, Li40 = out40
    g40m, pi40m, Li40m, al40, il40 = ref.VBx(fea, Phi, pi=int(kw['pi']), gamma=qinit, maxIters=40,
                                             epsilon=1e-6, loopProb=kw['loopProb'], Fa=kw['Fa'],
                                             Fb=kw['Fb'], return_model=True)
    assert np.array_equal(g40, g40m)
    g10, pi10, Li10, al10, il10 = ref.VBx(fea, Phi, pi=int(kw['pi']), gamma=qinit, maxIters=10,
                                          epsilon=1e-4, loopProb=kw['loopProb'], Fa=kw['Fa'],
                                          Fb=kw['Fb'], return_model=True)
    seg = np.loadtxt(f'{REF}/exp/ES2005a.seg', dtype=object)
    committed = parse_rttm(open(f'{REF}/exp/ES2005a.rttm').read())
    produced = parse_rttm(rttm_text)
    np.savez_compressed(
        os.path.join(HERE, 'es2005a.npz'),
        fea=fea, Phi=Phi, qinit=qinit, loopProb=kw['loopProb'], Fa=kw['Fa'], Fb=kw['Fb'],
        gamma40=g40, pi40=pi40, Li40=li_array(Li40), alpha40=al40, invL40=il40,
        gamma10=g10, pi10=pi10, Li10=li_array(Li10), alpha10=al10, invL10=il10,
        seg_times=seg[:, 2:].astype(np.float64),
        rttm_committed=committed, rttm_produced=produced)
    print('ES2005a: T,D,S =', fea.shape, qinit.shape[1], 'iters40 =', len(Li40),
          'final ELBO', Li40[-1][0], 'rttm segs', len(produced), len(committed))

    # ---------------------------------------------------------------- synthetic cases
    cases = {}
    GEN = {}

    def gen(T, S, seed, kappa):
        X, Phi, lab = make_recording(T, S, seed=seed, kappa=kappa)
        GEN[id(X)] = (T, S, seed, kappa)
        return X, Phi, lab

    def record(name, X, Phi, kw, seed_before=None):
        if seed_before is not None:
            np.random.seed(seed_before)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = ref.VBx(X.copy(), Phi.copy(), return_model=True, **kw)
        g, p, Li, al, il = out
        # X/Phi are regenerated from the generator arguments by the tests; a checksum
        # guards against generator drift.
        cases[name + '/gen'] = np.asarray(GEN[id(X)], dtype=np.float64)
        cases[name + '/X_checksum'] = np.asarray([X.sum(), (X ** 2).sum(), Phi.sum()])
        for k, v in kw.items():
            if v is not None:
                cases[name + '/kw_' + k] = np.asarray(v)
        if seed_before is not None:
            cases[name + '/np_seed'] = np.asarray(seed_before)
        cases[name + '/gamma'] = g
        cases[name + '/pi'] = p
        cases[name + '/Li'] = li_array(Li)
        if al is not None:          # maxIters=0 returns the (None) inputs, VBx.py:126
            cases[name + '/alpha'] = al
            cases[name + '/invL'] = il
        cases[name + '/warned'] = np.asarray('WARNING' in buf.getvalue())
        print(f'{name}: T={X.shape[0]} S={g.shape[1]} iters={len(Li)} warned={bool(cases[name + "/warned"])}')

    def soft_init(T, S, seed):
        r = np.random.default_rng(seed)
        q = r.gamma(1.0, size=(T, S))
        return q / q.sum(1, keepdims=True)

    X, Phi, _ = gen(600, 12, 3, 0.05)
    record('soft_T600_S12', X, Phi, dict(loopProb=0.9, Fa=0.3, Fb=17, pi=12, gamma=soft_init(600, 12, 11),
                                         maxIters=12, epsilon=-1e300))
    X, Phi, _ = gen(1000, 30, 4, 0.05)
    record('soft_T1000_S30', X, Phi, dict(loopProb=0.99, Fa=0.3, Fb=17, pi=30, gamma=soft_init(1000, 30, 12),
                                          maxIters=10, epsilon=-1e300))
    X, Phi, _ = gen(700, 50, 5, 0.1)
    record('soft_T700_S50', X, Phi, dict(loopProb=0.9, Fa=0.2, Fb=6, pi=50, gamma=soft_init(700, 50, 13),
                                         maxIters=8, epsilon=-1e300))
    X, Phi, _ = gen(500, 10, 6, 1.0)
    record('easy_T500_S10', X, Phi, dict(loopProb=0.99, Fa=0.3, Fb=17, pi=10, gamma=soft_init(500, 10, 14),
                                         maxIters=6, epsilon=-1e300))
    X, Phi, _ = gen(400, 10, 7, 0.05)
    record('rng_init_T400_S10', X, Phi, dict(loopProb=0.9, Fa=0.4, Fb=17, pi=10, gamma=None, maxIters=8,
                                             epsilon=1e-4, alphaQInit=1.0), seed_before=1)
    pivec = np.array([0.5, 0.0, 0.2, 0.0, 0.3, 0.0])
    X, Phi, _ = gen(300, 6, 8, 0.1)
    record('pi_vector_zeros_T300_S6', X, Phi, dict(loopProb=0.65, Fa=0.4, Fb=64, pi=pivec,
                                                   gamma=soft_init(300, 6, 15), maxIters=6, epsilon=-1e300))
    # warm start: alpha/invL from a previous run skip the first M-step (VBx.py:94)
    X, Phi, _ = gen(350, 8, 9, 0.1)
    g0 = soft_init(350, 8, 16)
    _, _, _, al0, il0 = ref.VBx(X, Phi, loopProb=0.9, Fa=0.3, Fb=17, pi=8, gamma=g0, maxIters=3,
                                epsilon=-1e300, return_model=True)
    record('warm_start_T350_S8', X, Phi, dict(loopProb=0.9, Fa=0.3, Fb=17, pi=8, gamma=g0, maxIters=4,
                                              epsilon=-1e300, alpha=al0, invL=il0))
    X, Phi, _ = gen(200, 1, 10, 0.1)
    record('single_speaker_T200_S1', X, Phi, dict(loopProb=0.9, Fa=0.3, Fb=17, pi=1, gamma=np.ones((200, 1)),
                                                  maxIters=10, epsilon=1e-4))
    X, Phi, _ = gen(1, 4, 11, 0.1)
    record('one_frame_S4', X, Phi, dict(loopProb=0.9, Fa=0.3, Fb=17, pi=4, gamma=soft_init(1, 4, 17),
                                        maxIters=3, epsilon=-1e300))
    X, Phi, _ = gen(2, 4, 12, 0.1)
    record('two_frames_S4', X, Phi, dict(loopProb=0.9, Fa=0.3, Fb=17, pi=4, gamma=soft_init(2, 4, 18),
                                         maxIters=3, epsilon=-1e300))
    X, Phi, _ = gen(250, 5, 13, 0.1)
    record('loop0_T250_S5', X, Phi, dict(loopProb=0.0, Fa=0.3, Fb=17, pi=5, gamma=soft_init(250, 5, 19),
                                         maxIters=5, epsilon=-1e300))
    record('loop1_T250_S5', X, Phi, dict(loopProb=1.0, Fa=0.3, Fb=17, pi=5, gamma=soft_init(250, 5, 19),
                                         maxIters=5, epsilon=-1e300))
    record('zero_iters_T250_S5', X, Phi, dict(loopProb=0.9, Fa=0.3, Fb=17, pi=5, gamma=soft_init(250, 5, 19),
                                              maxIters=0, epsilon=1e-4))
    # early stop with the default epsilon
    X, Phi, _ = gen(450, 9, 14, 0.3)
    record('early_stop_T450_S9', X, Phi, dict(loopProb=0.9, Fa=0.3, Fb=17, pi=9, gamma=soft_init(450, 9, 20),
                                              maxIters=30, epsilon=1e-4))
    # S > 64 (more states than lanes in a wavefront)
    X, Phi, _ = gen(300, 70, 15, 0.1)
    record('wide_T300_S70', X, Phi, dict(loopProb=0.9, Fa=0.3, Fb=17, pi=70, gamma=soft_init(300, 70, 21),
                                         maxIters=5, epsilon=-1e300))
    np.savez_compressed(os.path.join(HERE, 'synth_cases.npz'), **cases)

    # ---------------------------------------------------------------- forward_backward KATs
    fb = {}
    for name, (T, S, lp, seed) in {'fb_T64_S5': (64, 5, 0.9, 0), 'fb_T257_S31': (257, 31, 0.99, 1),
                                   'fb_T100_S64': (100, 64, 0.35, 2), 'fb_T3_S2': (3, 2, 0.5, 3),
                                   'fb_T1_S7': (1, 7, 0.9, 4), 'fb_T90_S3_lp1': (90, 3, 1.0, 5),
                                   'fb_T90_S3_lp0': (90, 3, 0.0, 6)}.items():
        lls, pi = make_lls(T, S, seed=seed)
        tr = np.eye(S) * lp + (1 - lp) * pi
        post, tll, lfw, lbw = ref.forward_backward(lls, tr, pi)
        fb[name + '/lls'] = lls
        fb[name + '/pi'] = pi
        fb[name + '/loopProb'] = np.asarray(lp)
        fb[name + '/post'] = post
        fb[name + '/tll'] = np.asarray(tll)
        fb[name + '/lfw'] = lfw
        fb[name + '/lbw'] = lbw
    np.savez_compressed(os.path.join(HERE, 'fb_cases.npz'), **fb)
    print('wrote', sorted(os.listdir(HERE)))


if __name__ == '__main__':
    main()
