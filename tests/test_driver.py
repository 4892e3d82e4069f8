"""The batched diarization driver ``vbx_amd.vbhmm`` and the on-disk formats either
side of the path (``vbx_amd.kaldi_formats``, ``vbx_amd.h5_minimal``).

Golden: tests/golden/driver_split3.npz -- RTTM files written by the UNCHANGED reference driver (its own VBx.py) for
a three-recording archive cut from the reference's example x-vectors (tests/golden/make_golden_driver.py).  The
tests rewrite the archive, the segments file and the model files with this repository's writers, run the driver and
compare the RTTM files byte for byte:

  * ``-m "not gpu"``: score stage and VB-HMM replaced by the CPU oracles (injected; the product has no fallback)
    -- formats, grouping, projections, AHC, batching, label post-processing, RTTM text; also on two ``gloo`` ranks;
  * ``-m gpu``: the product path end to end through ``vbx_amd.vbhmm.main``.
"""
import io
import os
import socket
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, 'tests', 'golden', 'driver_split3.npz')
REF = os.environ.get('VBX_REFERENCE', '/root/reference')
RECS = ('recA', 'recB', 'recC')


def _write_inputs(tmp):
    """The golden archive as files: Kaldi ark + segments + Kaldi PLDA + transform (npz: no HDF5 writer here)."""
    from vbx_amd import kaldi_formats as kf
    g = np.load(GOLD)
    tmp = str(tmp)
    paths = dict(ark=os.path.join(tmp, 'split3.ark'), seg=os.path.join(tmp, 'split3.seg'),
                 plda=os.path.join(tmp, 'plda'), transform=os.path.join(tmp, 'transform.npz'),
                 out=os.path.join(tmp, 'rttm'))
    kf.write_vec_flt_ark(paths['ark'], zip(g['keys'].tolist(), g['xvecs']))
    kf.write_segments(paths['seg'], [(k, r, s, e) for k, r, (s, e) in zip(g['keys'].tolist(), g['recs'].tolist(), g['segments'])])
    kf.write_plda(paths['plda'], g['plda_mean'], g['plda_trans'], g['plda_psi'])
    np.savez(paths['transform'], mean1=g['mean1'], mean2=g['mean2'], lda=g['lda'])
    return paths


def _argv(paths, extra=()):
    return ['--init', 'AHC+VB', '--out-rttm-dir', paths['out'], '--xvec-ark-file', paths['ark'],
            '--segments-file', paths['seg'], '--xvec-transform', paths['transform'], '--plda-file', paths['plda'],
            '--threshold', '-0.015', '--lda-dim', '128', '--Fa', '0.3', '--Fb', '17', '--loopP', '0.99',
            '--output-2nd', 'True'] + list(extra)


def _check_rttm(paths, recs=RECS):
    g = np.load(GOLD)
    for rec in recs:
        assert open(os.path.join(paths['out'], rec + '.rttm'), 'rb').read() == g['rttm_' + rec].tobytes(), rec
        assert open(os.path.join(paths['out'] + '2nd', rec + '.rttm'), 'rb').read() == g['rttm2_' + rec].tobytes(), rec


class OracleStages:
    """The four stage methods of vbx_amd.vbhmm.DeviceStages on the CPU checkers (oracle/, NumPy restatements of
    vbhmm.py:125-129,150-153,160-162): what stands in for the GPU in the CPU test-suite."""

    def __init__(self):
        self.vb_calls = 0

    def project(self, recordings, models, lda_dim):
        from vbx_amd.vbhmm import l2_norm
        self.x = [l2_norm(models['lda'].T.dot(l2_norm(r[2] - models['mean1']).transpose()).transpose() - models['mean2'])
                  for r in recordings]                                                        # vbhmm.py:129
        self.fea = [(x - models['plda_mu']).dot(models['plda_tr'].T)[:, :lda_dim] for x in self.x]   # vbhmm.py:153
        self.Phi = models['plda_psi'][:lda_dim]

    def ahc(self, k, threshold):
        from oracle import ahc_oracle
        from scipy.cluster.hierarchy import fcluster, linkage
        from scipy.spatial.distance import squareform
        m = ahc_oracle.cos_similarity(self.x[k])
        thr, _ = ahc_oracle.twoGMMcalib_lin(m.ravel())
        lin_mat = linkage(squareform(-m, checks=False), method='average')                     # vbhmm.py:139-141
        adjust = abs(lin_mat[:, 2].min())
        lin_mat[:, 2] += adjust
        return fcluster(lin_mat, -(thr + threshold) + adjust, criterion='distance') - 1, float(thr)   # vbhmm.py:142-146

    def vb(self, ks, labels, init_smoothing, maxIters, epsilon, precision, **hyper):
        from oracle import vbx_oracle                               # checker standing in for the GPU
        from scipy.special import softmax
        self.vb_calls += 1
        out = []
        for k, lab in zip(ks, labels):
            qinit = np.zeros((len(lab), np.max(lab) + 1))
            qinit[range(len(lab)), lab] = 1.0
            qinit = softmax(qinit * init_smoothing, axis=1)                                   # vbhmm.py:150-152
            q, _sp, L = vbx_oracle.VBx(self.fea[k], self.Phi, pi=qinit.shape[1], gamma=qinit, maxIters=maxIters,
                                       epsilon=epsilon, **hyper)
            order = np.argsort(-q, axis=1)                                                    # vbhmm.py:160-162
            out.append((order[:, 0], order[:, 1] if q.shape[1] > 1 else None, len(L)))
        return out

    def close(self):
        pass


# ---- formats ----------------------------------------------------------------------------------------------------
def test_vector_archive_round_trip_binary_and_text(tmp_path):
    from vbx_amd import kaldi_formats as kf
    rng = np.random.default_rng(0)
    items = [(f'rec{k // 3}_{k:04d}-a', rng.standard_normal(7 + k).astype(np.float32)) for k in range(6)]
    for dtype in (np.float32, np.float64):
        p = str(tmp_path / f'v_{np.dtype(dtype).name}.ark')
        kf.write_vec_flt_ark(p, items, dtype=dtype)
        back = list(kf.read_vec_flt_ark(p))
        assert [k for k, _ in back] == [k for k, _ in items]
        for (_, a), (_, b) in zip(items, back):
            assert b.dtype == np.dtype(dtype) and np.array_equal(a.astype(dtype), b)
    # Kaldi text archive: 'key  [ v v v ]\n'
    p = str(tmp_path / 't.ark')
    with open(p, 'w') as fd:
        fd.write('utt_1  [ 1.5 -2 3e-1 ]\nutt_2  [ ]\nutt_3  [ 7 ]\n')
    back = list(kf.read_vec_flt_ark(p))
    assert [k for k, _ in back] == ['utt_1', 'utt_2', 'utt_3']
    assert np.allclose(back[0][1], [1.5, -2, 0.3]) and back[1][1].size == 0 and np.allclose(back[2][1], [7])
    with open(p, 'wb') as fd:
        fd.write(b'utt_1 \x00BXV \x04\x01\x00\x00\x00')
    with pytest.raises(ValueError):
        list(kf.read_vec_flt_ark(p))


def test_plda_round_trip_and_text_form(tmp_path):
    from vbx_amd import kaldi_formats as kf
    rng = np.random.default_rng(1)
    mean, trans, psi = rng.standard_normal(5), rng.standard_normal((5, 5)), rng.uniform(0.5, 6, 5)
    for dtype in (np.float32, np.float64):
        p = str(tmp_path / f'plda_{np.dtype(dtype).name}')
        kf.write_plda(p, mean, trans, psi, dtype=dtype)
        m, t, s = kf.read_plda(p)
        assert t.shape == (5, 5) and m.dtype == np.dtype(dtype)
        assert np.array_equal(m, mean.astype(dtype)) and np.array_equal(t, trans.astype(dtype)) and np.array_equal(s, psi.astype(dtype))
    p = str(tmp_path / 'plda_text')
    with open(p, 'w') as fd:          # what `ivector-copy-plda --binary=false` prints
        fd.write('<Plda>  [ 1 2 ]\n [\n  1 0.5 \n  0 2 ]\n [ 3 4 ]\n</Plda> ')
    m, t, s = kf.read_plda(p)
    assert np.array_equal(m, [1, 2]) and np.array_equal(t, [[1, 0.5], [0, 2]]) and np.array_equal(s, [3, 4])
    with open(p, 'wb') as fd:
        fd.write(b'\x00B<Nnet> ')
    with pytest.raises(ValueError):
        kf.read_plda(p)


def test_segments_rttm_and_label_merging(tmp_path):
    from vbx_amd import kaldi_formats as kf
    from vbx_amd.vbhmm import merge_adjacent_labels
    p = str(tmp_path / 'seg')
    kf.write_segments(p, [('a_0', 'a', 0.0, 1.44), ('a_1', 'a', 0.24, 1.68), ('b_0', 'b', 0.0, 1.0)])
    d = kf.read_xvector_timing_dict(p)
    assert list(d) == ['a', 'b'] and d['a'][0].tolist() == ['a_0', 'a_1'] and np.array_equal(d['a'][1], [[0, 1.44], [0.24, 1.68]])
    # same label: overlapping and touching segments merge; different labels: the overlap is split in the middle
    starts, ends, labels = merge_adjacent_labels(np.array([0.0, 0.24, 0.48, 2.0, 2.5]), np.array([1.44, 1.68, 1.92, 2.5, 3.0]),
                                                 np.array([3, 3, 1, 1, 1]))
    assert labels.tolist() == [3, 1, 1] and np.allclose(starts, [0.0, 1.08, 2.0]) and np.allclose(ends, [1.08, 1.92, 3.0])
    buf = io.StringIO()
    kf.write_rttm(buf, 'rec', labels, starts, ends)
    assert buf.getvalue().splitlines()[0] == 'SPEAKER rec 1 0.000000 1.080000 <NA> <NA> 4 <NA> <NA>'
    q = str(tmp_path / 'r.rttm')
    open(q, 'w').write(buf.getvalue())
    assert kf.read_rttm(q)[1] == ('rec', 1.08, 0.84, '2')


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, 'VBx', 'models')), reason='reference checkout not present')
def test_readers_on_the_reference_files():
    """The reference's own model files and example archive (authoring container): the HDF5 subset reader against
    the known layout of transform.h5 (three contiguous float64 datasets), the PLDA and archive readers against the
    values SURVEY.md App. B quotes."""
    from vbx_amd import kaldi_formats as kf
    from vbx_amd.h5_minimal import read_datasets
    path = f'{REF}/VBx/models/ResNet101_16kHz/transform.h5'
    d = read_datasets(path)
    raw = open(path, 'rb').read()
    for name, (off, shape) in {'mean1': (2048, (256,)), 'mean2': (4096, (128,)), 'lda': (5120, (256, 128))}.items():
        assert np.array_equal(d[name], np.frombuffer(raw, '<f8', int(np.prod(shape)), off).reshape(shape))
    with pytest.raises(KeyError):
        read_datasets(path, ('nope',))
    mean, trans, psi = kf.read_plda(f'{REF}/VBx/models/ResNet101_16kHz/plda')
    assert mean.shape == (128,) and trans.shape == (128, 128) and psi.shape == (128,)
    items = list(kf.read_vec_flt_ark(f'{REF}/exp/ES2005a.ark'))
    assert len(items) == 1025 and items[0][0] == 'ES2005a_0000-00000000-00000144' and items[0][1].shape == (256,)
    g = np.load(GOLD)
    assert np.array_equal(g['xvecs'], np.array([v for _, v in items])) and np.array_equal(g['plda_psi'], psi)


def test_h5_reader_rejects_what_it_does_not_handle(tmp_path):
    from vbx_amd.h5_minimal import read_datasets
    p = str(tmp_path / 'x.h5')
    open(p, 'wb').write(b'not hdf5 at all')
    with pytest.raises(ValueError):
        read_datasets(p)
    open(p, 'wb').write(b'\x89HDF\r\n\x1a\n' + bytes([2]) + bytes(64))      # superblock version 2
    with pytest.raises(ValueError, match='superblock version 2'):
        read_datasets(p)


# ---- the driver ---------------------------------------------------------------------------------------------------
def test_driver_reproduces_the_reference_rttm_with_oracle_stages(tmp_path, capsys):
    from vbx_amd import vbhmm
    paths = _write_inputs(tmp_path)
    args = vbhmm.build_parser().parse_args(_argv(paths))
    state, timing = vbhmm.diarize(args, stages=OracleStages())
    assert capsys.readouterr().out.split() == list(RECS)                 # vbhmm.py:121 prints each recording
    assert list(state) == list(RECS) and timing['recordings'] == 3 and timing['xvectors'] == 1025
    assert all(st['n_iters'] >= 2 for st in state.values())
    _check_rttm(paths)


def test_driver_ahc_only_and_argument_checks(tmp_path):
    from vbx_amd import vbhmm
    paths = _write_inputs(tmp_path)
    argv = _argv(paths)
    argv[1] = 'AHC'
    args = vbhmm.build_parser().parse_args(argv)
    stages = OracleStages()
    state, _ = vbhmm.diarize(args, stages=stages, log=lambda *_: None)
    assert not stages.vb_calls and all(st['n_iters'] == 0 and st['labels2nd'] is None for st in state.values())
    assert sorted(os.listdir(paths['out'])) == [r + '.rttm' for r in RECS] and not os.path.exists(paths['out'] + '2nd')
    args.loopP = 1.5
    with pytest.raises(AssertionError):                                    # vbhmm.py:103
        vbhmm.diarize(args, stages=OracleStages(), log=lambda *_: None)
    with pytest.raises(SystemExit):
        vbhmm.build_parser().parse_args(['--init', 'VB'])


def test_driver_leaves_out_only_the_recording_it_cannot_take(tmp_path):
    """A recording whose VB-HMM stage fails (here: one the stage object refuses) costs only itself: the others are
    diarized and written, then the error names it."""
    from vbx_amd import vbhmm
    from vbx_amd._capi import VbxError
    paths = _write_inputs(tmp_path)
    args = vbhmm.build_parser().parse_args(_argv(paths))

    class Picky(OracleStages):
        def vb(self, ks, labels, *a, **k):
            if 1 in ks:                                   # recB: in the batch, and again alone
                raise VbxError('recording refused')
            return super().vb(ks, labels, *a, **k)
    with pytest.raises(RuntimeError, match='recB'):
        vbhmm.diarize(args, stages=Picky(), log=lambda *_: None)
    assert sorted(os.listdir(paths['out'])) == ['recA.rttm', 'recC.rttm']
    want = np.load(GOLD)
    for rec in ('recA', 'recC'):
        assert open(os.path.join(paths['out'], rec + '.rttm'), 'rb').read() == want['rttm_' + rec].tobytes()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    import torch.distributed as dist
    from vbx_amd import vbhmm
    import test_driver as td
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        paths = {k: os.path.join(tmp, v) for k, v in dict(ark='split3.ark', seg='split3.seg', plda='plda',
                                                          transform='transform.npz', out='rttm').items()}
        args = vbhmm.build_parser().parse_args(td._argv(paths))
        state, timing = vbhmm.diarize(args, stages=td.OracleStages(), log=lambda *_: None)
        assert (timing['rank'], timing['world']) == (rank, world)
        with open(os.path.join(tmp, f'rank{rank}.txt'), 'w') as fd:
            fd.write(' '.join(state))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_driver_on_two_gloo_ranks_writes_every_recording_once(tmp_path):
    """One process per GPU under torchrun: recordings are dealt to the ranks (longest first, by T^2), each rank
    writes the RTTM files of its own recordings, nothing is exchanged."""
    import torch.multiprocessing as mp
    paths = _write_inputs(tmp_path)
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    owned = [open(tmp_path / f'rank{r}.txt').read().split() for r in range(world)]
    assert sorted(owned[0] + owned[1]) == list(RECS) and owned[0] and owned[1]
    assert owned[0] == ['recA']                       # 400^2 > 325^2 + 300^2 is false: LPT gives rank 0 recA, rank 1 recC + recB
    _check_rttm(paths)


@pytest.mark.gpu
def test_driver_reproduces_the_reference_rttm_on_the_gpu(tmp_path, capsys):
    """Product path: GPU score stage + one fp64 ``vbx_batch`` for the three recordings, through the command line."""
    from vbx_amd import vbhmm
    paths = _write_inputs(tmp_path)
    assert vbhmm.main(_argv(paths, ['--timing'])) == 0
    out = capsys.readouterr().out.split('\n')
    assert out[:3] == list(RECS)
    _check_rttm(paths)
    # fp32 kernels: same segments and speakers on this example (labels are arg-max decisions)
    paths32 = dict(paths, out=os.path.join(str(tmp_path), 'rttm32'))
    assert vbhmm.main(_argv(paths32, ['--precision', 'fp32'])) == 0
    _check_rttm(paths32)


@pytest.mark.gpu
def test_driver_takes_a_recording_with_more_than_256_ahc_clusters(tmp_path):
    """AHC on a long or noisy file can leave VBx() hundreds of clusters (vbhmm.py:150-158 hands it one HMM state per
    cluster) and the reference takes any number (VBx.py:76-85).  Round 2 reported and skipped such a recording; now it
    runs (its own batch: the driver buckets recordings by padded state count) and the other recordings are untouched."""
    from vbx_amd import vbhmm
    from oracle import vbx_oracle
    from scipy.special import softmax
    paths = _write_inputs(tmp_path)
    args = vbhmm.build_parser().parse_args(_argv(paths))

    class Many(vbhmm.DeviceStages):
        def ahc(self, k, threshold):
            labels, thr = super().ahc(k, threshold)
            if k == 1:                                        # recB (300 x-vectors): 290 clusters
                labels = np.arange(len(labels)) % 290
                self.fea_b = self.xv.get('fea', self.row0[k], self.T[k])
            return labels, thr
    stages = Many()
    state, timing = vbhmm.diarize(args, stages=stages, log=lambda *_: None)
    assert list(state) == list(RECS) and timing['recordings'] == 3
    _check_rttm(paths, recs=('recA', 'recC'))
    lab = np.arange(300) % 290
    qinit = np.zeros((300, 290))
    qinit[range(300), lab] = 1.0
    qinit = softmax(qinit * args.init_smoothing, axis=1)
    q, _, L = vbx_oracle.VBx(stages.fea_b, stages.Phi, pi=290, gamma=qinit, maxIters=40, epsilon=1e-6, loopProb=args.loopP,
                             Fa=args.Fa, Fb=args.Fb)
    assert state['recB']['n_iters'] == len(L)
    assert np.array_equal(state['recB']['labels1st'], np.argsort(-q, axis=1, kind='stable')[:, 0])


@pytest.mark.gpu
def test_device_stages_against_what_the_reference_driver_computed(tmp_path):
    """The steps either side of VBx() on the device against the values captured from the UNCHANGED reference driver
    (tests/golden/frontend_split3.npz, make_golden_frontend.py): projected x-vectors (vbhmm.py:125-129), PLDA
    projection (vbhmm.py:153), first / second speaker of every x-vector after the batched VB-HMM started from the
    reference's AHC labels (vbhmm.py:160-162), and the initial responsibilities built on the device from those labels
    (vbhmm.py:150-152), checked through one EM iteration against the oracle started from the reference's own qinit."""
    from vbx_amd import vbhmm, _capi
    from oracle import vbx_oracle
    paths = _write_inputs(tmp_path)
    args = vbhmm.build_parser().parse_args(_argv(paths))
    models = vbhmm.load_models(args.xvec_transform, args.plda_file)
    recordings = vbhmm._read_recordings(args.xvec_ark_file)
    ref = np.load(os.path.join(os.path.dirname(GOLD), 'frontend_split3.npz'))
    stages = vbhmm.DeviceStages()
    stages.project(recordings, models, args.lda_dim)
    ks, labels = [], []
    for k, rec in enumerate(RECS):
        T = len(recordings[k][1])
        xproj = stages.xv.get('xproj', stages.row0[k], T)
        fea = stages.xv.get('fea', stages.row0[k], T)
        np.testing.assert_allclose(xproj, ref[rec + '/xproj'], rtol=0, atol=1e-14, err_msg=rec)
        np.testing.assert_allclose(fea, ref[rec + '/fea'], rtol=0, atol=1e-11, err_msg=rec)     # (|fea| up to ~10; the mean goes through the product)
        ks.append(k)
        labels.append(np.argmax(ref[rec + '/qinit'], axis=1))            # the reference's AHC labels
    # device qinit + batch + device arg-sort == the reference's VBx() call on ITS fea / qinit, label by label
    out = stages.vb(ks, labels, init_smoothing=args.init_smoothing, maxIters=40, epsilon=1e-6, precision='fp64',
                    loopProb=args.loopP, Fa=args.Fa, Fb=args.Fb)
    for rec, (first, second, n_iters) in zip(RECS, out):
        assert np.array_equal(first, ref[rec + '/labels1st']), rec
        assert np.array_equal(second, ref[rec + '/labels2nd']), rec
    # one iteration from the device-built initial responsibilities == the oracle from the reference's qinit
    batch = _capi.Batch(stages.ctx, [stages.T[0]], [ref['recA/qinit'].shape[1]], args.lda_dim, precision='fp64', max_iters=1)
    batch.set_recording_resident(0, stages.xv, 0, labels[0], args.init_smoothing, stages.Phi, args.loopP, args.Fa, args.Fb)
    batch.run(1, -np.inf)
    g, p, L = vbx_oracle.VBx(ref['recA/fea'], ref['recA/Phi'], pi=ref['recA/qinit'].shape[1], gamma=ref['recA/qinit'], maxIters=1,
                             epsilon=-1e300, loopProb=args.loopP, Fa=args.Fa, Fb=args.Fb)
    res = batch.result(0, want_model=False)
    batch.close()
    stages.close()
    np.testing.assert_allclose(res['gamma'], g, rtol=0, atol=1e-9)
    np.testing.assert_allclose(res['Li'], [L[0][0]], rtol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['cosine', 'ties', 'es2005a'])
def test_device_linkage_is_the_host_linkage_bit_for_bit(kind, monkeypatch):
    """vbx_scores_linkage_average in its CHAIN form (VBX_AMD_LINKAGE_DEVICE=chain: the nearest-neighbour chain walked by one
    workgroup on the score matrix in HBM) against vbx_linkage_average on the condensed matrix the device hands out (itself
    SciPy's linkage bit for bit, see below): same merges, same numbering, same distances to the last bit -- random cosine
    similarities, similarities with massive exact ties (few distinct integer-valued x-vectors), the example recording."""
    from vbx_amd import _capi
    monkeypatch.setenv('VBX_AMD_EXPERIMENT', '1')
    monkeypatch.setenv('VBX_AMD_LINKAGE_DEVICE', 'chain')
    ctx = _capi.default_context()
    rng = np.random.default_rng(11)
    xs = []
    if kind == 'cosine':
        # (from 8192 clusters the chain runs in stages with the live rows / columns compacted in between)
        xs = [rng.standard_normal((n, 16)) for n in (2, 3, 5, 64, 257, 1000, 2500, 8192, 9001)]
    elif kind == 'ties':
        for n in (7, 40, 300, 1200, 8300):
            base = rng.integers(-1, 2, size=(6, 8)).astype(float) + 0.5           # six distinct directions: duplicates galore
            xs.append(base[rng.integers(0, 6, n)])
    else:
        xs = [np.load(GOLD)['xvecs'][:400].astype(np.float64), np.load(GOLD)['xvecs'].astype(np.float64)]
    for x in xs:
        n = len(x)
        sc = _capi.Scores.cos_similarity(ctx, x)
        cond = sc.get_condensed(n, -1.0)
        got = sc.linkage_average(n)
        sc.close()
        want = _capi.linkage_average(cond)
        assert got.shape == want.shape and np.array_equal(got, want), (kind, n, np.argwhere(got != want)[:3])
        assert np.array_equal(np.signbit(got), np.signbit(want))


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['cosine', 'clustered', 'ties', 'es2005a'])
def test_device_linkage_in_rounds_of_reciprocal_pairs(kind, monkeypatch):
    """The default device linkage from 1024 clusters on: ALL reciprocal nearest-neighbour pairs merged per round by the whole
    chip (vbx_ahc.hpp, rnn_*), the chain finishing the last few hundred clusters.  Average linkage is reducible, so the
    merges are the chain's; only the order in which a cluster's updates meet differs.  Against the chain form (= the host
    routine = SciPy, bit for bit):
      * distinct distances (cosine, clustered speakers, the example recording): the SAME tree -- identical children and
        sizes in every row of Z -- and heights within 16 ulp (measured: 8); identical flat clusters at any threshold;
      * massive exact ties: a valid average-linkage dendrogram of the input (every height = the mean pairwise distance of
        the two merged clusters), and identical flat clusters at thresholds that do not fall on a tied height."""
    from vbx_amd import _capi
    ctx = _capi.default_context()
    rng = np.random.default_rng(17)
    xs = []
    if kind == 'cosine':
        xs = [rng.standard_normal((n, 16)) for n in (1024, 1500, 4097, 9001)]
    elif kind == 'clustered':
        for n, k in ((3000, 6), (10000, 9)):                                        # speakers: tight groups, like real x-vectors
            centres = rng.standard_normal((k, 32))
            xs.append(centres[rng.integers(0, k, n)] + 0.35 * rng.standard_normal((n, 32)))
    elif kind == 'ties':
        for n in (1200, 2500):
            base = rng.integers(-1, 2, size=(6, 8)).astype(float) + 0.5
            xs.append(base[rng.integers(0, 6, n)])
    else:
        xv = np.load(GOLD)['xvecs'].astype(np.float64)
        xs = [xv, np.concatenate([xv, xv[::-1] * 1.0001 + 1e-3 * rng.standard_normal(xv.shape)])]    # 1025, 2050
    for x in xs:
        n = len(x)
        monkeypatch.setenv('VBX_AMD_EXPERIMENT', '1')
        monkeypatch.setenv('VBX_AMD_LINKAGE_DEVICE', 'rounds')
        sc = _capi.Scores.cos_similarity(ctx, x)
        cond = sc.get_condensed(n, -1.0)
        got = sc.linkage_average(n)
        sc.close()
        monkeypatch.setenv('VBX_AMD_LINKAGE_DEVICE', 'chain')
        sc = _capi.Scores.cos_similarity(ctx, x)
        want = sc.linkage_average(n)
        sc.close()
        assert got.shape == want.shape == (n - 1, 4)
        assert np.all(np.diff(got[:, 2]) >= 0) and np.array_equal(got[:, 3][-1:], [n])
        if kind != 'ties':
            assert np.array_equal(got[:, [0, 1, 3]], want[:, [0, 1, 3]]), (kind, n, np.argwhere(got[:, [0, 1, 3]] != want[:, [0, 1, 3]])[:3])
            ulp = np.spacing(np.abs(want[:, 2]))
            # (a height is a chain of weighted means whose association order differs between the two forms: measured <= 8 ulp)
            # (... of the height, or of the distances it is a mean of where positive and negative ones cancel: |d| <= 1)
            assert np.all(np.abs(got[:, 2] - want[:, 2]) <= np.maximum(16 * ulp, 1e-15)), (kind, n, np.max(np.abs(got[:, 2] - want[:, 2]) / ulp))
            cuts = np.quantile(want[:, 2], [0.2, 0.5, 0.8, 0.95, 0.99])
        else:
            if n <= 1500:
                _upgma_check(cond, got)
            levels = np.unique(np.round(want[:, 2], 9))
            cuts = (levels[:-1] + levels[1:]) / 2                                    # between two tied heights
            cuts = cuts[:: max(1, len(cuts) // 6)]
        for t in cuts:
            a, b = _capi.fcluster_distance(got, float(t)), _capi.fcluster_distance(want, float(t))
            # the same partition (cluster numbers follow the tree walk, which ties may order differently)
            assert len(set(zip(a.tolist(), b.tolist()))) == len(set(a.tolist())) == len(set(b.tolist())), (kind, n, t)


def _upgma_check(y, Z):
    """Z is a valid average-linkage dendrogram of the condensed distances y: every merge distance is the mean pairwise
    distance between the two merged clusters (to rounding), heights do not decrease, sizes add up."""
    from scipy.spatial.distance import squareform
    n = len(Z) + 1
    full = squareform(y) if n > 2 else np.array([[0.0, y[0]], [y[0], 0.0]])
    members = {i: [i] for i in range(n)}
    for k, (a, b, d, m) in enumerate(Z):
        A, B = members.pop(int(a)), members.pop(int(b))
        assert a < b and m == len(A) + len(B)
        np.testing.assert_allclose(d, full[np.ix_(A, B)].mean(), rtol=1e-12, atol=1e-13)
        members[n + k] = A + B
    assert np.all(np.diff(Z[:, 2]) >= -1e-13)


@pytest.mark.parametrize('case', ['cosine', 'real', 'ties', 'near_ties'])
def test_fastcluster_form_of_the_average_linkage(case):
    """vbhmm.py:140-141 calls fastcluster.linkage; every fixture here was made with SciPy's linkage because fastcluster is
    not installed.  ``linkage_average(y, 'fastcluster')`` restates the package's own routine (weights divided before the
    update, its own chain bookkeeping: vbx_linkage.hpp).  Without ties: the same tree as SciPy's form, distances equal
    to a few units in the last place, the same flat clusters at any threshold that is not within rounding of a merge
    height.  With exact or near ties either form may pick another, equally valid, merge: both are checked to be average
    linkage dendrograms of the input."""
    from vbx_amd import _capi
    from scipy.cluster.hierarchy import fcluster
    rng = np.random.default_rng(11)
    ys = []
    if case == 'cosine':
        for n in (2, 3, 4, 9, 64, 257, 700):
            x = rng.standard_normal((n, 12))
            x /= np.linalg.norm(x, axis=1, keepdims=True)
            ys.append(np.ascontiguousarray(-(x @ x.T)[np.triu_indices(n, 1)]))
    elif case == 'real':
        x = np.load(GOLD)['xvecs'].astype(np.float64)         # the example recording, raw x-vectors
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        ys = [np.ascontiguousarray(-(x[:400] @ x[:400].T)[np.triu_indices(400, 1)]),
              np.ascontiguousarray(-(x @ x.T)[np.triu_indices(len(x), 1)])]
    elif case == 'ties':
        ys = [rng.integers(0, 4, size=n * (n - 1) // 2).astype(float) for n in (5, 6, 33, 150)]
    else:
        for n in (30, 200):                                    # distances that differ in their last bits only
            base = rng.integers(1, 5, size=n * (n - 1) // 2).astype(float)
            ys.append(base * (1.0 + rng.integers(-2, 3, size=base.size) * 2.0 ** -52))
    for y in ys:
        n = int(round((1 + np.sqrt(1 + 8 * y.size)) / 2))
        Zs, Zf = _capi.linkage_average(y, 'scipy'), _capi.linkage_average(y, 'fastcluster')
        shift = abs(y.min()) + 1.0                             # (the UPGMA check wants non-negative distances)
        _upgma_check(y + shift, np.column_stack([Zs[:, :2], Zs[:, 2] + shift, Zs[:, 3]]))
        _upgma_check(y + shift, np.column_stack([Zf[:, :2], Zf[:, 2] + shift, Zf[:, 3]]))
        if case in ('cosine', 'real'):
            assert np.array_equal(Zs[:, [0, 1, 3]], Zf[:, [0, 1, 3]]), (case, n)
            assert np.abs(Zs[:, 2] - Zf[:, 2]).max() <= 4 * np.finfo(float).eps * max(1.0, np.abs(Zs[:, 2]).max())
            if n > 2:
                Zs2, Zf2 = Zs.copy(), Zf.copy()                  # (SciPy's fcluster wants non-negative heights: vbhmm.py:142-144)
                Zs2[:, 2] += shift
                Zf2[:, 2] += shift
                for t in np.quantile(Zs2[:, 2], [0.3, 0.7, 0.95]) + 1e-9:
                    assert np.array_equal(fcluster(Zs2, t, 'distance'), fcluster(Zf2, t, 'distance'))
    with pytest.raises(ValueError):
        _capi.linkage_average(np.zeros(3), 'ward')


def test_both_linkage_forms_give_the_committed_example_clustering(monkeypatch):
    """The one artefact made with the REAL fastcluster is the reference's exp/ES2005a.rttm (es2005a.npz: rttm_committed
    equals what the SciPy stand-in produced, up to the numbering of the speakers).  It cannot tell the two forms apart:
    on the example's score matrix both give the same tree and the same 31 clusters at the calibrated threshold."""
    from vbx_amd import vbhmm
    from oracle import ahc_oracle
    ahc = np.load(os.path.join(os.path.dirname(GOLD), 'ahc_cases.npz'))
    es = np.load(os.path.join(os.path.dirname(GOLD), 'es2005a.npz'))
    x = ahc['es2005a/x']
    scr = ahc_oracle.cos_similarity(x)
    cond = np.ascontiguousarray(-scr[np.triu_indices(len(x), 1)])
    labels = {}
    for variant in ('scipy', 'fastcluster'):
        monkeypatch.setenv('VBX_AMD_LINKAGE', variant)
        labels[variant] = vbhmm.cluster(cond.copy(), float(ahc['es2005a/thr']), -0.015)
    assert np.array_equal(labels['scipy'], labels['fastcluster'])
    assert labels['scipy'].max() + 1 == es['qinit'].shape[1] == 31
    assert np.array_equal(labels['scipy'], np.argmax(es['qinit'], axis=1))      # the labels the reference driver went on with
    same = (es['rttm_committed'][:, :2] == es['rttm_produced'][:, :2]).all()
    assert same and len(set(zip(es['rttm_committed'][:, 2], es['rttm_produced'][:, 2]))) == len(set(es['rttm_committed'][:, 2]))


def test_top2_ties_keep_their_index_order():
    """(host-side statement of the tie rule the device arg-sort follows: a stable sort of -q)"""
    q = np.array([[0.5, 0.5, 0.0], [0.2, 0.4, 0.4], [0.0, 0.0, 1.0]])
    order = np.argsort(-q, axis=1, kind='stable')
    assert order[:, 0].tolist() == [0, 1, 2] and order[:, 1].tolist() == [1, 2, 0]


# ---- native average linkage (host code of the library: no GPU needed) -------------------------------------------
@pytest.mark.parametrize('case', ['cosine', 'ties', 'negative_ties', 'zeros', 'es2005a'])
def test_native_average_linkage_is_scipy_linkage_bit_for_bit(case):
    """vbx_linkage_average == scipy.cluster.hierarchy.linkage(y, 'average') (the stand-in for fastcluster.linkage of
    vbhmm.py:140-141; SciPy 1.15 `nn_chain` + `label`): same merges, same cluster numbering, same distances to the
    last bit, sign of zero included -- also with massive ties, where the chain's tie rules decide the tree."""
    from scipy.cluster.hierarchy import linkage
    from vbx_amd import _capi
    rng = np.random.default_rng(5)
    ys = []
    if case == 'cosine':
        for n in (2, 3, 4, 9, 64, 257, 700):
            x = rng.standard_normal((n, 12))
            x /= np.linalg.norm(x, axis=1, keepdims=True)
            ys.append(np.ascontiguousarray(-(x @ x.T)[np.triu_indices(n, 1)]))
    elif case == 'ties':
        ys = [rng.integers(0, 4, size=n * (n - 1) // 2).astype(float) for n in (5, 6, 33, 150, 301)]
    elif case == 'negative_ties':
        ys = [-rng.integers(0, 3, size=n * (n - 1) // 2).astype(float) for n in (5, 40, 300)]       # has -0.0
    elif case == 'zeros':
        ys = [np.zeros(n * (n - 1) // 2) for n in (2, 10, 65)]
    else:
        g = np.load(GOLD)                                     # the example recording's first 400 x-vectors, raw
        x = g['xvecs'][:400].astype(np.float64)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        ys = [np.ascontiguousarray(-(x @ x.T)[np.triu_indices(400, 1)])]
    for y in ys:
        want = linkage(y, method='average')
        got = _capi.linkage_average(y)
        assert got.shape == want.shape and np.array_equal(got, want), (case, len(want) + 1)
        assert np.array_equal(np.signbit(got), np.signbit(want))
    with pytest.raises(ValueError):
        _capi.linkage_average(np.zeros(4))                    # 4 is not n (n - 1) / 2
    with pytest.raises(ValueError):
        _capi.linkage_average(np.zeros((3, 3)))


def test_native_linkage_runs_concurrently_from_threads():
    """The entry point keeps no global state and ctypes drops the interpreter lock around it: four recordings
    clustered on four threads give the results of four calls in a row."""
    from concurrent.futures import ThreadPoolExecutor
    from vbx_amd import _capi
    rng = np.random.default_rng(6)
    ys = []
    for n in (300, 301, 450, 200):
        x = rng.standard_normal((n, 8))
        ys.append(np.ascontiguousarray(-(x @ x.T)[np.triu_indices(n, 1)]))
    serial = [_capi.linkage_average(y) for y in ys]
    with ThreadPoolExecutor(4) as pool:
        threaded = list(pool.map(_capi.linkage_average, ys * 3))
    for k, z in enumerate(threaded):
        assert np.array_equal(z, serial[k % 4])


def test_native_distance_cut_is_scipy_fcluster():
    """vbx_fcluster_distance == scipy.cluster.hierarchy.fcluster(Z, t, criterion='distance') (vbhmm.py:145-146): same
    members, same numbering, for cuts below the first merge, above the last, at exact merge distances and with ties."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from vbx_amd import _capi
    rng = np.random.default_rng(8)
    for n in (2, 3, 5, 17, 100, 401):
        x = rng.standard_normal((n, 8))
        Z = _capi.linkage_average(np.ascontiguousarray(-(x @ x.T)[np.triu_indices(n, 1)]))
        Z[:, 2] += abs(Z[:, 2].min())                           # vbhmm.py:143-144
        for t in list(np.quantile(Z[:, 2], [0, 0.1, 0.5, 0.9, 1.0])) + [-1.0, 1e9] + list(Z[::7, 2]):
            got = _capi.fcluster_distance(Z, t)
            assert got.dtype == np.int32 and np.array_equal(got, fcluster(Z, t, criterion='distance')), (n, t)
    Z = linkage(rng.integers(0, 3, size=60 * 59 // 2).astype(float), 'average')
    for t in (0, 0.5, 1, 1.5, 2, 3):
        assert np.array_equal(_capi.fcluster_distance(Z, t), fcluster(Z, t, criterion='distance')), t
    with pytest.raises(_capi.VbxError):                          # child id out of range
        _capi.fcluster_distance(np.array([[0.0, 5.0, 1.0, 2.0], [1.0, 2.0, 2.0, 3.0]]), 1.0)
    with pytest.raises(ValueError):
        _capi.fcluster_distance(np.zeros((3, 3)), 1.0)


def test_grouped_archive_reader_equals_the_entry_by_entry_reader(tmp_path):
    """read_vec_flt_ark_grouped (native index + one gather per recording) == itertools.groupby over read_vec_flt_ark
    (vbhmm.py:117-123), for float32 and float64 archives, keys of different lengths, a one-vector recording; a text
    archive takes the fallback; a truncated archive is refused."""
    import itertools
    from vbx_amd import kaldi_formats as kf
    rng = np.random.default_rng(9)
    items = []
    for rec, n in (('meeting-A', 5), ('b', 1), ('rec_with_under_scores', 7), ('NA', 2)):
        items += [(f'{rec}_{k:03d}-x', rng.standard_normal(16).astype(np.float32)) for k in range(n)]
    for dtype in (np.float32, np.float64):
        p = str(tmp_path / f'g_{np.dtype(dtype).name}.ark')
        kf.write_vec_flt_ark(p, items, dtype=dtype)
        got = kf.read_vec_flt_ark_grouped(p)
        want = [(name, list(g)) for name, g in itertools.groupby(kf.read_vec_flt_ark(p), lambda e: e[0].rsplit('_', 1)[0])]
        assert [g[0] for g in got] == [w[0] for w in want] == ['meeting-A', 'b', 'rec_with_under_scores', 'NA']
        for (name, keys, mat), (_, entries) in zip(got, want):
            assert keys.tolist() == [k for k, _ in entries] and mat.dtype == np.dtype(dtype)
            assert np.array_equal(mat, np.array([v for _, v in entries]))
    raw = open(p, 'rb').read()
    open(p, 'wb').write(raw[:-5])                                # last vector cut short
    with pytest.raises(ValueError):
        kf.read_vec_flt_ark_grouped(p)
    t = str(tmp_path / 't.ark')
    open(t, 'w').write('r_1  [ 1 2 ]\nr_2  [ 3 4 ]\ns_1  [ 5 6 ]\n')
    got = kf.read_vec_flt_ark_grouped(t)
    assert [g[0] for g in got] == ['r', 's'] and np.array_equal(got[0][2], [[1, 2], [3, 4]])
    seg = str(tmp_path / 'seg')
    kf.write_segments(seg, [('NA_000-x', 'NA', 0.0, 1.0), ('nan_1', 'nan', 1.0, 2.5)])      # names a CSV parser may eat
    d = kf.read_xvector_timing_dict(seg)
    assert list(d) == ['NA', 'nan'] and d['NA'][0].tolist() == ['NA_000-x'] and np.array_equal(d['nan'][1], [[1.0, 2.5]])


def test_native_clustering_randomised_against_scipy():
    """400 random condensed matrices (Gaussian, small-integer, one-decimal and cosine distances: from no ties to almost
    only ties), n = 2 .. 119: linkage matrix and distance cut equal SciPy's in every case."""
    from scipy.cluster.hierarchy import fcluster, linkage
    from vbx_amd import _capi
    rng = np.random.default_rng(123)
    for trial in range(400):
        n = int(rng.integers(2, 120))
        m = n * (n - 1) // 2
        kind = trial % 4
        if kind == 0:
            y = rng.standard_normal(m)
        elif kind == 1:
            y = rng.integers(0, 3, m).astype(float)
        elif kind == 2:
            y = np.round(rng.standard_normal(m), 1)
        else:
            x = rng.standard_normal((n, 3))
            x /= np.linalg.norm(x, axis=1, keepdims=True)
            y = np.ascontiguousarray(-(x @ x.T)[np.triu_indices(n, 1)])
        want = linkage(y, 'average')
        got = _capi.linkage_average(y)
        assert np.array_equal(got, want) and np.array_equal(np.signbit(got), np.signbit(want)), (trial, n, kind)
        want[:, 2] += abs(want[:, 2].min())
        t = float(rng.choice(want[:, 2])) if rng.random() < 0.7 else float(abs(rng.standard_normal()))
        assert np.array_equal(_capi.fcluster_distance(want, t), fcluster(want, t, criterion='distance')), (trial, t)
