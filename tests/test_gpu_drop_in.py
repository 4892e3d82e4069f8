"""One run, on the GPU, of the path the UNCHANGED reference driver takes into this repository (-m gpu):

    tools/run_vbhmm.py  ->  <checkout>/VBx/vbhmm.py  --  ``from VBx import VBx`` (vbhmm.py:45)  ->  vbx_drop_in/VBx.py
                        ->  vbx_amd.VBx  ->  ctypes  ->  libvbx_hip.so  ->  HIP kernels

The GPU box has no checkout of the reference, so the "checkout" is the stand-in tests/make_mini_vbhmm.py emits: a minimal
caller with the driver's own import lines and its call of VBx() in the shape of vbhmm.py:154-158, fed from the committed
ES2005a fixtures.  (Where the reference exists, tests/test_drop_in_launcher.py runs the real vbhmm.py through the same
launcher -- without a GPU.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def test_minimal_caller_with_the_drivers_imports_reaches_the_kernels(tmp_path, es2005a):
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import make_mini_vbhmm
    make_mini_vbhmm.emit(str(tmp_path / 'checkout'))
    out = tmp_path / 'rttm'
    out.mkdir()
    stale = ('VBx', 'diarization_lib', '_reference_diarization_lib', 'kaldi_utils', 'run_vbhmm')
    for name in stale:
        sys.modules.pop(name, None)
    try:
        import run_vbhmm
        run_vbhmm.main(['--reference', str(tmp_path / 'checkout'), '--allow-shims', '--',
                        '--fixture', os.path.join(GOLDEN, 'es2005a.npz'), '--ahc-fixture', os.path.join(GOLDEN, 'ahc_cases.npz'),
                        '--out-rttm-dir', str(out), '--lda-dim', '128', '--Fa', '0.3', '--Fb', '17', '--loopP', '0.99'])
        drop_in = sys.modules['VBx']
        assert os.path.samefile(drop_in.__file__, os.path.join(REPO, 'vbx_drop_in', 'VBx.py'))
        dlib = sys.modules['diarization_lib']
        assert os.path.samefile(dlib.__file__, os.path.join(REPO, 'vbx_drop_in', 'diarization_lib.py'))
        import vbx_amd.diarization_lib
        assert dlib.cos_similarity is vbx_amd.diarization_lib.cos_similarity       # the two redirected functions ...
        assert callable(dlib.merge_adjacent_labels) and callable(dlib.mkdir_p)     # ... and the checkout's own names
    finally:
        for name in stale:
            sys.modules.pop(name, None)
        sys.path.remove(os.path.join(REPO, 'tools'))
    from vbx_amd import _capi
    assert _capi._lib is not None and os.path.samefile(_capi.library_path(), os.path.join(REPO, 'vbx_amd', 'csrc', 'libvbx_hip.so'))
    ahc = np.load(os.path.join(GOLDEN, 'ahc_cases.npz'))
    np.testing.assert_allclose(np.load(out / 'thr.npy'), ahc['es2005a/thr'], rtol=1e-10)     # vbhmm.py:135-138 on the device
    assert int(np.load(out / 'n_iters.npy')) == len(es2005a['Li40']) == 13                    # the reference's own stop
    rows = []
    for line in open(out / 'ES2005a.rttm'):
        f = line.split()
        rows.append((float(f[3]), float(f[4]), int(f[7])))
    assert np.array_equal(np.array(rows), es2005a['rttm_produced'])
