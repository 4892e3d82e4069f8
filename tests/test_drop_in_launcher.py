"""The unchanged reference driver vbhmm.py, launched by tools/run_vbhmm.py, imports this
repository's VBx module (vbhmm.py:45) and calls it exactly as vbhmm.py:154-158 does.

Runs only where a checkout of the reference exists (the authoring container); the device call is
replaced by the CPU oracle here because the container has no GPU -- what is under test is the
import redirection and the call contract, not the kernels (tests/test_gpu_parity.py::
test_es2005a_end_to_end covers the same call on the MI355X from the committed fixture)."""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('VBX_REFERENCE', '/root/reference')


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'VBx', 'vbhmm.py')), reason='reference checkout not present')
def test_unchanged_vbhmm_runs_on_the_drop_in_module(tmp_path, monkeypatch, es2005a):
    from oracle import vbx_oracle
    import vbx_amd                                              # noqa: F401
    product = sys.modules['vbx_amd.VBx']                        # the submodule (vbx_amd.VBx the attribute is the function)
    calls = []

    def recording_vbx(X, Phi, **kw):
        calls.append((np.array(X), np.array(Phi), dict(kw)))
        return vbx_oracle.VBx(X, Phi, **kw)

    monkeypatch.setattr(product, 'VBx', recording_vbx)
    # the AHC score stage is redirected too (vbx_drop_in/diarization_lib.py): oracle stand-ins, calls recorded
    from oracle import ahc_oracle
    import vbx_amd.diarization_lib as product_ahc
    ahc_calls = []

    def recording_cos(x):
        ahc_calls.append(('cos_similarity', np.shape(x)))
        return ahc_oracle.cos_similarity(x)

    def recording_gmm(s, niters=20):
        ahc_calls.append(('twoGMMcalib_lin', np.shape(s)))
        return ahc_oracle.twoGMMcalib_lin(s, niters)

    monkeypatch.setattr(product_ahc, 'cos_similarity', recording_cos)
    monkeypatch.setattr(product_ahc, 'twoGMMcalib_lin', recording_gmm)
    for name in ('VBx', 'kaldi_io', 'kaldi_io.kaldi_io', 'h5py', 'fastcluster', 'diarization_lib', 'kaldi_utils'):
        monkeypatch.delitem(sys.modules, name, raising=False)
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    try:
        import run_vbhmm
        run_vbhmm.main(['--reference', REF, '--allow-shims', '--',
                        '--init', 'AHC+VB', '--out-rttm-dir', str(tmp_path),
                        '--xvec-ark-file', f'{REF}/exp/ES2005a.ark', '--segments-file', f'{REF}/exp/ES2005a.seg',
                        '--xvec-transform', f'{REF}/VBx/models/ResNet101_16kHz/transform.h5',
                        '--plda-file', f'{REF}/VBx/models/ResNet101_16kHz/plda',
                        '--threshold', '-0.015', '--lda-dim', '128', '--Fa', '0.3', '--Fb', '17', '--loopP', '0.99'])
        drop_in = sys.modules['VBx']
        assert os.path.samefile(drop_in.__file__, os.path.join(REPO, 'vbx_drop_in', 'VBx.py'))
        dlib = sys.modules['diarization_lib']
        assert os.path.samefile(dlib.__file__, os.path.join(REPO, 'vbx_drop_in', 'diarization_lib.py'))
        assert callable(dlib.merge_adjacent_labels) and callable(dlib.read_xvector_timing_dict)   # reference names
    finally:
        sys.path.remove(os.path.join(REPO, 'tools'))
        for name in ('VBx', 'run_vbhmm', 'kaldi_io', 'kaldi_io.kaldi_io', 'h5py', 'fastcluster', 'diarization_lib',
                     '_reference_diarization_lib', 'kaldi_utils'):
            sys.modules.pop(name, None)
    assert ahc_calls == [('cos_similarity', (1025, 128)), ('twoGMMcalib_lin', (1025 * 1025,))]   # vbhmm.py:135-138
    assert len(calls) == 1
    X, Phi, kw = calls[0]
    assert np.array_equal(X, es2005a['fea']) and np.array_equal(Phi, es2005a['Phi'])
    assert np.array_equal(kw['gamma'], es2005a['qinit'])
    assert (kw['pi'], kw['maxIters'], kw['epsilon']) == (31, 40, 1e-6)             # vbhmm.py:154-158
    rows = []
    for line in open(tmp_path / 'ES2005a.rttm'):
        f = line.split()
        rows.append((float(f[3]), float(f[4]), int(f[7])))
    assert np.array_equal(np.array(rows), es2005a['rttm_produced'])


def test_generated_minimal_caller_goes_through_the_same_redirection(tmp_path, monkeypatch, es2005a):
    """tests/make_mini_vbhmm.py (what the -m gpu test tests/test_gpu_drop_in.py runs on the box without a reference
    checkout): same launcher, same import lines, same call shape -- here with the oracle behind the drop-in."""
    from oracle import vbx_oracle, ahc_oracle
    import vbx_amd                                              # noqa: F401
    import vbx_amd.diarization_lib as product_ahc
    product = sys.modules['vbx_amd.VBx']
    calls = []

    def recording_vbx(X, Phi, **kw):
        calls.append(dict(kw))
        return vbx_oracle.VBx(X, Phi, **kw)

    monkeypatch.setattr(product, 'VBx', recording_vbx)
    monkeypatch.setattr(product_ahc, 'cos_similarity', ahc_oracle.cos_similarity)
    monkeypatch.setattr(product_ahc, 'twoGMMcalib_lin', ahc_oracle.twoGMMcalib_lin)
    stale = ('VBx', 'diarization_lib', '_reference_diarization_lib', 'kaldi_utils', 'run_vbhmm')
    for name in stale:
        monkeypatch.delitem(sys.modules, name, raising=False)
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    out = tmp_path / 'rttm'
    out.mkdir()
    try:
        import make_mini_vbhmm
        import run_vbhmm
        make_mini_vbhmm.emit(str(tmp_path / 'checkout'))
        gold = os.path.join(REPO, 'tests', 'golden')
        run_vbhmm.main(['--reference', str(tmp_path / 'checkout'), '--allow-shims', '--',
                        '--fixture', os.path.join(gold, 'es2005a.npz'), '--ahc-fixture', os.path.join(gold, 'ahc_cases.npz'),
                        '--out-rttm-dir', str(out), '--lda-dim', '128', '--Fa', '0.3', '--Fb', '17', '--loopP', '0.99'])
        assert os.path.samefile(sys.modules['VBx'].__file__, os.path.join(REPO, 'vbx_drop_in', 'VBx.py'))
        assert os.path.samefile(sys.modules['diarization_lib'].__file__, os.path.join(REPO, 'vbx_drop_in', 'diarization_lib.py'))
    finally:
        sys.path.remove(os.path.join(REPO, 'tools'))
        sys.path.remove(os.path.join(REPO, 'tests'))
        for name in stale:
            sys.modules.pop(name, None)
    assert len(calls) == 1 and (calls[0]['pi'], calls[0]['maxIters'], calls[0]['epsilon']) == (31, 40, 1e-6)    # vbhmm.py:154-158
    assert np.array_equal(calls[0]['gamma'], es2005a['qinit'])
    rows = [(float(f[3]), float(f[4]), int(f[7])) for f in (line.split() for line in open(out / 'ES2005a.rttm'))]
    assert np.array_equal(np.array(rows), es2005a['rttm_produced'])
    assert int(np.load(out / 'n_iters.npy')) == 13
