"""Generator of a stand-in for a checkout of the reference, for the box that has none (-m gpu tests run where
/root/reference does not exist): ``emit(root)`` writes

    <root>/VBx/vbhmm.py            a MINIMAL caller with the unchanged driver's import lines (vbhmm.py:41-45) and its one call
                                   of VBx() in exactly the shape of vbhmm.py:154-158, around the score stage of vbhmm.py:135-138
                                   and the label post-processing of vbhmm.py:160-179; inputs come from the committed fixtures
    <root>/VBx/diarization_lib.py  the "reference" module the drop-in loads from the next sys.path entry and re-exports
    <root>/VBx/kaldi_utils.py      (this repository's implementations under the reference's names)

so that ``tools/run_vbhmm.py --reference <root>`` exercises, on the GPU, the very redirection the unchanged driver goes
through: ``from VBx import VBx`` must resolve to vbx_drop_in/VBx.py, ``from diarization_lib import ...`` to
vbx_drop_in/diarization_lib.py, and the call must reach the HIP kernels.  (tests/test_drop_in_launcher.py runs the real,
unchanged vbhmm.py where the reference exists -- without a GPU, with the oracle behind the drop-in.)
"""
import os
import textwrap

CALLER = '''
    import argparse
    import os

    import numpy as np
    from scipy.special import softmax

    from diarization_lib import read_xvector_timing_dict, l2_norm, \\
        cos_similarity, twoGMMcalib_lin, merge_adjacent_labels, mkdir_p
    from kaldi_utils import read_plda
    from VBx import VBx

    parser = argparse.ArgumentParser()
    parser.add_argument('--fixture', required=True)
    parser.add_argument('--ahc-fixture', required=True)
    parser.add_argument('--out-rttm-dir', required=True)
    parser.add_argument('--lda-dim', required=True, type=int)
    parser.add_argument('--Fa', required=True, type=float)
    parser.add_argument('--Fb', required=True, type=float)
    parser.add_argument('--loopP', required=True, type=float)
    parser.add_argument('--init-smoothing', required=False, type=float, default=5.0)
    args = parser.parse_args()

    fix = np.load(args.fixture)
    file_name = 'ES2005a'
    x = np.load(args.ahc_fixture)['es2005a/x']            # the projected x-vectors of vbhmm.py:129
    scr_mx = cos_similarity(x)
    thr, _ = twoGMMcalib_lin(scr_mx.ravel())
    np.save(os.path.join(args.out_rttm_dir, 'thr.npy'), np.array(thr))
    # (the clustering between vbhmm.py:139 and :146 is host code of the reference's dependencies: its labels come from
    #  the fixture, smoothed exactly as vbhmm.py:150-152 does)
    labels1st = np.argmax(fix['qinit'], axis=1)
    qinit = np.zeros((len(labels1st), np.max(labels1st) + 1))
    qinit[range(len(labels1st)), labels1st] = 1.0
    qinit = softmax(qinit * args.init_smoothing, axis=1)
    fea, plda_psi = fix['fea'], fix['Phi']
    q, sp, L = VBx(
        fea, plda_psi[:args.lda_dim],
        pi=qinit.shape[1], gamma=qinit,
        maxIters=40, epsilon=1e-6,
        loopProb=args.loopP, Fa=args.Fa, Fb=args.Fb)

    labels1st = np.argsort(-q, axis=1)[:, 0]
    start, end = fix['seg_times'].T
    starts, ends, out_labels = merge_adjacent_labels(start, end, labels1st)
    mkdir_p(args.out_rttm_dir)
    with open(os.path.join(args.out_rttm_dir, f'{file_name}.rttm'), 'w') as fp:
        for label, seg_start, seg_end in zip(out_labels, starts, ends):
            fp.write(f'SPEAKER {file_name} 1 {seg_start:03f} {seg_end - seg_start:03f} '
                     f'<NA> <NA> {label + 1} <NA> <NA>{os.linesep}')
    np.save(os.path.join(args.out_rttm_dir, 'n_iters.npy'), np.array(len(L)))
'''

DIARIZATION_LIB = '''
    """Stand-in for the reference's diarization_lib.py (names of vbhmm.py:41-42), on this repository's implementations."""
    import errno
    import os

    from vbx_amd.vbhmm import l2_norm, merge_adjacent_labels          # noqa: F401
    from vbx_amd.kaldi_formats import read_xvector_timing_dict         # noqa: F401


    def mkdir_p(path):
        try:
            os.makedirs(path)
        except OSError as exc:
            if exc.errno != errno.EEXIST or not os.path.isdir(path):
                raise


    def cos_similarity(x):
        raise RuntimeError('the reference implementation was called: the drop-in did not take over')


    def twoGMMcalib_lin(s, niters=20):
        raise RuntimeError('the reference implementation was called: the drop-in did not take over')
'''

KALDI_UTILS = '''
    from vbx_amd.kaldi_formats import read_plda                        # noqa: F401
'''

VBX_DECOY = '''
    def VBx(*args, **kwargs):
        raise RuntimeError('the sibling VBx.py of the driver was imported: the drop-in did not take over')
'''


def emit(root):
    """Write the stand-in checkout under ``root``; returns the path of its vbhmm.py."""
    d = os.path.join(root, 'VBx')
    os.makedirs(d, exist_ok=True)
    for name, src in (('vbhmm.py', CALLER), ('diarization_lib.py', DIARIZATION_LIB), ('kaldi_utils.py', KALDI_UTILS),
                      ('VBx.py', VBX_DECOY)):          # (a sibling VBx.py, as in the reference: it must lose to the drop-in)
        with open(os.path.join(d, name), 'w') as f:
            f.write(textwrap.dedent(src).lstrip('\n'))
    return os.path.join(d, 'vbhmm.py')
