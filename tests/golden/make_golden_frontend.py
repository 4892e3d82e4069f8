#!/usr/bin/env python
"""What the reference driver computes EITHER SIDE of VBx() -- the x-vector projection (vbhmm.py:125-129), the initial
soft assignments (vbhmm.py:150-152), the PLDA projection ``fea`` (vbhmm.py:153) and the first / second speaker labels
(vbhmm.py:160-162) -- captured while the UNCHANGED /root/reference/VBx/vbhmm.py runs over the three-recording
archive of driver_split3.npz (authoring container only).  Two recording shims sit in front of the reference's own
modules: ``diarization_lib`` (everything from the reference; ``cos_similarity`` notes its argument = the projected
x-vectors) and ``VBx`` (the reference's VBx(); notes its inputs and outputs).

    tests/golden/frontend_split3.npz   per recording: xproj [T][128], fea [T][128], qinit [T][S], q [T][S],
                                       labels1st, labels2nd (np.argsort(-q, axis=1)[:, 0 / 1] as vbhmm.py:160-162)
"""
import os
import runpy
import sys
import tempfile
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))

from vbx_amd import kaldi_formats as kf              # noqa: E402
import run_vbhmm                                     # noqa: E402

RECORDERS = {
    'VBx.py': '''
        import importlib.util, numpy as np
        _spec = importlib.util.spec_from_file_location('_ref_VBx', '%(ref)s/VBx/VBx.py')
        _ref = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_ref)
        CALLS = []
        def VBx(X, Phi, **kw):
            out = _ref.VBx(X, Phi, **kw)
            CALLS.append((np.array(X), np.array(Phi), {k: np.array(v) for k, v in kw.items()}, out))
            return out
    ''' % {'ref': REF},
    'diarization_lib.py': '''
        import importlib.util, numpy as np
        _spec = importlib.util.spec_from_file_location('_ref_dl', '%(ref)s/VBx/diarization_lib.py')
        _ref = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(_ref)
        globals().update({k: v for k, v in vars(_ref).items() if not k.startswith('__')})
        PROJECTED = []
        def cos_similarity(x):
            PROJECTED.append(np.array(x))
            return _ref.cos_similarity(x)
    ''' % {'ref': REF},
}


def main():
    g = np.load(os.path.join(HERE, 'driver_split3.npz'))
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        ark, seg = os.path.join(tmp, 'split3.ark'), os.path.join(tmp, 'split3.seg')
        kf.write_vec_flt_ark(ark, zip(g['keys'], g['xvecs']))
        kf.write_segments(seg, [(k, r, s, e) for k, r, (s, e) in zip(g['keys'], g['recs'], g['segments'])])
        shims = os.path.join(tmp, 'shims')
        run_vbhmm.write_stand_ins(shims, list(run_vbhmm.STAND_INS))
        for rel, src in RECORDERS.items():
            with open(os.path.join(shims, rel), 'w') as f:
                f.write(textwrap.dedent(src))
        argv = ['--init', 'AHC+VB', '--out-rttm-dir', os.path.join(tmp, 'rttm'), '--xvec-ark-file', ark,
                '--segments-file', seg, '--xvec-transform', f'{REF}/VBx/models/ResNet101_16kHz/transform.h5',
                '--plda-file', f'{REF}/VBx/models/ResNet101_16kHz/plda', '--threshold', '-0.015', '--lda-dim', '128',
                '--Fa', '0.3', '--Fb', '17', '--loopP', '0.99', '--output-2nd', 'True']
        script = f'{REF}/VBx/vbhmm.py'
        old_argv, old_path = sys.argv, list(sys.path)
        sys.argv = [script] + argv
        sys.path[:0] = [shims, f'{REF}/VBx']
        for name in ('VBx', 'diarization_lib', 'kaldi_utils', 'kaldi_io', 'h5py', 'fastcluster'):
            sys.modules.pop(name, None)
        try:
            runpy.run_path(script, run_name='__main__')
            calls = sys.modules['VBx'].CALLS
            projected = sys.modules['diarization_lib'].PROJECTED
        finally:
            sys.argv, sys.path[:] = old_argv, old_path
    assert len(calls) == len(projected) == 3
    for rec, x, (fea, Phi, kw, (q, sp, L)) in zip(('recA', 'recB', 'recC'), projected, calls):
        out[rec + '/xproj'] = x
        out[rec + '/fea'] = fea
        out[rec + '/Phi'] = Phi
        out[rec + '/qinit'] = kw['gamma']
        out[rec + '/q'] = q
        out[rec + '/labels1st'] = np.argsort(-q, axis=1)[:, 0]                    # vbhmm.py:160
        if q.shape[1] > 1:
            out[rec + '/labels2nd'] = np.argsort(-q, axis=1)[:, 1]                # vbhmm.py:162
        srt = -np.sort(-q, axis=1)
        ties = int(np.sum(srt[:, 1] == srt[:, 2])) if q.shape[1] > 2 else 0
        print(rec, x.shape, fea.shape, q.shape, len(L), 'iterations; rows with a tie for the second place:', ties)
    np.savez_compressed(os.path.join(HERE, 'frontend_split3.npz'), **out)


if __name__ == '__main__':
    main()
