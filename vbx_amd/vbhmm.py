#!/usr/bin/env python
"""Diarization driver with the command line of the reference's ``VBx/vbhmm.py``, built around the batched
MI355X path:

    python -m vbx_amd.vbhmm --init AHC+VB --out-rttm-dir out --xvec-ark-file x.ark --segments-file x.seg \\
        --xvec-transform transform.h5 --plda-file plda --threshold -0.015 --lda-dim 128 --Fa 0.3 --Fb 17 --loopP 0.99

The reference walks the recordings of the archive one after another (vbhmm.py:120) and runs everything on one
CPU thread.  Here the loop is split into stages:

  0. ALL x-vectors of this rank go to the device once; the projections of vbhmm.py:125-129 and the PLDA projection
     ``fea`` of vbhmm.py:153 are computed there for the whole archive and stay resident;
  1. per recording: AHC initialisation -- similarity matrix of the resident rows, threshold calibration and the
     average-linkage nearest-neighbour chain on the device (one persistent workgroup per recording, several recordings
     side by side on their own streams), the cut of the dendrogram on the host;
  2. ALL recordings of this rank in one ``vbx_batch``: initial responsibilities from the AHC labels on the device
     (vbhmm.py:150-152), one launch sequence per EM iteration for the whole archive, convergence per recording on the
     device (vbhmm.py:154-158 call, batched), first / second speaker by a device arg-sort (vbhmm.py:160-162);
  3. per recording, as soon as its labels exist: merging of adjacent segments and the RTTM file (vbhmm.py:166-179).

Under ``torchrun`` (one process per GPU) the recordings are dealt to the ranks by cost; every rank writes the
RTTM files of its own recordings and the ranks only meet in a barrier at the end.  Same flags, same files, same
RTTM bytes as the reference driver.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

from .kaldi_formats import (read_plda, read_vec_flt_ark_grouped, read_xvec_transform, read_xvector_timing_dict,
                            write_rttm)

__all__ = ['main', 'diarize', 'DeviceStages', 'tune_host_process', 'load_models', 'cluster', 'cut_linkage', 'merge_adjacent_labels', 'l2_norm']


def l2_norm(vec_or_matrix):
    """Rows (or the vector) scaled to unit Euclidean length.  diarization_lib.py:169-187."""
    a = np.asarray(vec_or_matrix)
    if a.ndim == 1:
        return a / np.linalg.norm(a)
    if a.ndim == 2:
        return a / np.linalg.norm(a, axis=1, ord=2)[:, np.newaxis]
    raise ValueError('Wrong number of dimensions, 1 or 2 is supported, not %i.' % a.ndim)


def merge_adjacent_labels(starts, ends, labels):
    """Adjacent or overlapping segments with the same label become one; overlapping segments with different
    labels meet in the middle of the overlap.  diarization_lib.py:116-137."""
    starts, ends, labels = np.asarray(starts), np.asarray(ends), np.asarray(labels)
    touching = np.logical_or(np.isclose(ends[:-1], starts[1:]), ends[:-1] > starts[1:])
    cut = np.nonzero(np.logical_or(~touching, labels[1:] != labels[:-1]))[0]
    starts = starts[np.r_[0, cut + 1]]
    ends = ends[np.r_[cut, -1]]
    labels = labels[np.r_[0, cut + 1]]
    over = np.nonzero(starts[1:] < ends[:-1])[0]
    ends[over] = starts[over + 1] = (ends[over] + starts[over + 1]) / 2.0
    return starts, ends, labels


def load_models(xvec_transform, plda_file):
    """x-vector transform and the PLDA model in the simultaneously diagonalised form of vbhmm.py:107-113."""
    from scipy.linalg import eigh
    mean1, mean2, lda = read_xvec_transform(xvec_transform)
    plda_mu, plda_tr, plda_psi = read_plda(plda_file)
    with _small_blas_pool():       # 256 x 256: 0.30 s with the BLAS pool of a 256-core host, 0.01 s on four threads
        W = np.linalg.inv(plda_tr.T.dot(plda_tr))
        B = np.linalg.inv((plda_tr.T / plda_psi).dot(plda_tr))
        acvar, wccn = eigh(B, W)
    return dict(mean1=mean1, mean2=mean2, lda=lda, plda_mu=plda_mu, plda_psi=acvar[::-1], plda_tr=wccn.T[::-1])


def cut_linkage(lin_mat, thr, threshold):
    """vbhmm.py:142-146: the linkage matrix shifted to non-negative distances and cut at the calibrated threshold ->
    integer cluster label per x-vector (``vbx_fcluster_distance``: SciPy's fcluster(criterion='distance'), native)."""
    from . import _capi
    if len(lin_mat) == 0:
        return np.zeros(1, dtype=np.int64)
    adjust = abs(lin_mat[:, 2].min())
    lin_mat[:, 2] += adjust
    return _capi.fcluster_distance(lin_mat, -(thr + threshold) + adjust).astype(np.int64) - 1


def cluster(cond, thr, threshold):
    """Average-linkage clustering of condensed negated similarities held on the HOST, cut at the calibrated threshold
    (vbhmm.py:140-146): ``vbx_linkage_average`` (nearest-neighbour chain, SciPy's linkage matrix bit for bit -- SciPy is
    the stand-in for the un-installed fastcluster, whose average-linkage update may differ from it in the last bit).
    The driver itself clusters on the device (``DeviceStages.ahc``); this serves callers with their own matrix."""
    from . import _capi
    return cut_linkage(_capi.linkage_average(cond), thr, threshold)


class DeviceStages:
    """The device side of the driver (no host fallback: without the HIP library every method raises).

    project   every x-vector of this rank goes up ONCE; the projections of vbhmm.py:125-129 and the PLDA projection of
              vbhmm.py:153 run on the device for all recordings together and stay resident (vbx_xvectors)
    ahc       per recording: cosine-similarity matrix of its resident rows, two-Gaussian calibration and the
              average-linkage nearest-neighbour chain, all on the T x T matrix where it lies in HBM (vbhmm.py:135-141);
              the T - 1 merges come back and are cut on the host (vbhmm.py:142-146).  Every driver thread has its own
              context (device stream), so the chains of several recordings run side by side on different CUs
    vb        ALL recordings in one ``vbx_batch``: initial responsibilities built on the device from the AHC labels
              (vbhmm.py:150-152), one launch sequence per EM iteration for the whole archive, convergence per recording
              on the device (vbhmm.py:154-158), first / second speaker by a device arg-sort (vbhmm.py:160-162): only
              labels come back
    """

    def __init__(self, device=None):
        import threading
        from . import _capi
        self._capi = _capi
        self.ctx = _capi.default_context(device)
        self.xv = None
        self._local = threading.local()
        self._thread_ctxs = []

    def _thread_ctx(self):
        """A context (= a HIP stream) of its own for every driver thread: a ctx is not thread-safe."""
        ctx = getattr(self._local, 'ctx', None)
        if ctx is None:
            ctx = self._local.ctx = self._capi.Context(self.ctx.device)
            self._thread_ctxs.append(ctx)
        return ctx

    def project(self, recordings, models, lda_dim):
        self.T = [len(r[1]) for r in recordings]
        self.row0 = np.concatenate([[0], np.cumsum(self.T)]).astype(np.int64)
        self.Phi = np.ascontiguousarray(models['plda_psi'][:lda_dim])
        self.lda_dim = lda_dim
        if not recordings:
            return
        x = np.concatenate([np.asarray(r[2]) for r in recordings])
        self.xv = self._capi.XVectors(self.ctx, x, models['mean1'], models['lda'], models['mean2'], models['plda_mu'],
                                      models['plda_tr'], lda_dim)

    # The device linkage merges all reciprocal nearest-neighbour pairs per round on the whole chip (round 4: 1.2 ms at
    # T = 1025, 2.2 ms at 4000, 5.8 ms at 10 000, 19 ms at 20 000; the one-workgroup chain of rounds 2-3: 12 / 54 / 174 / 512 ms);
    # the host routine needs ~3 ns per matrix entry plus the condensed matrix over PCIe (3 ms at T = 1025): the device from ~600
    DEVICE_LINKAGE_FROM = int((os.environ.get('VBX_AMD_DEVICE_LINKAGE_FROM') if os.environ.get('VBX_AMD_EXPERIMENT') == '1' else None) or '600')

    def ahc(self, k, threshold):
        """-> (AHC labels of recording k, calibrated threshold)."""
        sc = self._capi.Scores.cos_similarity_resident(self._thread_ctx(), self.xv, self.row0[k], self.T[k])
        try:
            thr, _ = sc.two_gmm_calib(20, want_llr=False)
            if self.T[k] >= self.DEVICE_LINKAGE_FROM and self._capi.linkage_variant() == 'scipy':
                lin_mat = sc.linkage_average(self.T[k])                # (SciPy's update formula; the tree of the host routine)
            else:                                     # short recording: the host chain beats the device's step latency
                lin_mat = self._capi.linkage_average(sc.get_condensed(self.T[k], -1.0)) if self.T[k] > 1 else np.empty((0, 4))
        finally:
            sc.close()
        return cut_linkage(lin_mat, thr, threshold), float(thr)

    @staticmethod
    def padded_states(n_states):
        """The padded state count a batch of this many speakers runs with (vbx_host_batch.hpp: powers of two from 16)."""
        sp = 16
        while sp < n_states:
            sp *= 2
        return sp

    def vb(self, ks, labels, init_smoothing, maxIters, epsilon, precision, loopProb, Fa, Fb):
        """-> [(labels1st, labels2nd or None, iterations)] for the recordings ``ks`` with AHC labels ``labels``.

        One ``vbx_batch`` per padded state count: every recording of a batch runs with the widest one's padding, and one
        recording with more than 64 AHC clusters would push a whole archive from the fused kernels (responsibilities and
        lattices on the chip) onto the wide scan -- about 5x slower per iteration."""
        S = [int(np.max(lab)) + 1 for lab in labels]
        buckets = {}
        for j, s in enumerate(S):
            buckets.setdefault(self.padded_states(s), []).append(j)
        out = [None] * len(ks)
        for _sp, members in sorted(buckets.items()):
            batch = self._capi.Batch(self.ctx, [self.T[ks[j]] for j in members], [S[j] for j in members], self.lda_dim,
                                     precision=precision, max_iters=maxIters)
            try:
                for b, j in enumerate(members):
                    batch.set_recording_resident(b, self.xv, self.row0[ks[j]], labels[j], init_smoothing, self.Phi,
                                                 loopProb, Fa, Fb)
                batch.run(maxIters, epsilon)
                for b, j in enumerate(members):
                    first, second = batch.labels(b)
                    out[j] = (first, second, batch.n_iters(b))
            finally:
                batch.close()
        return out

    def close_thread_contexts(self):
        """The per-thread contexts (a HIP stream each) of the AHC stage: closed before the VB batch creates its stream
        group, so that the group's streams do not share hardware queues with idle ones (8 queues per process)."""
        for ctx in self._thread_ctxs:
            ctx.close()
        self._thread_ctxs = []
        self._local = __import__('threading').local()

    def close(self):
        if self.xv is not None:
            self.xv.close()
            self.xv = None
        self.close_thread_contexts()


def _small_blas_pool():
    """Context manager: at most four BLAS / LAPACK threads (no-op without ``threadpoolctl`` or with VBX_AMD_TUNE_HOST=0)."""
    import contextlib
    if os.environ.get('VBX_AMD_TUNE_HOST', '1') == '0':
        return contextlib.nullcontext()
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
    except ImportError:
        return contextlib.nullcontext()
    # never RAISE a pool: a BLAS that started with one thread (torchrun exports OMP_NUM_THREADS=1) has per-thread buffers
    # for one, and scipy.linalg.eigh on a pool widened to four crashed in LAPACK (SIGSEGV, seen on both ranks of a node)
    now = [i.get('num_threads', 1) for i in threadpool_info() if i.get('user_api') == 'blas']
    limit = min([4] + now)
    return threadpool_limits(limits=limit, user_api='blas')


def tune_host_process():
    """Process-wide settings for a run that mixes worker threads with many multi-megabyte NumPy temporaries (measured
    on a 256-core host: without them the per-recording host work of stage 1 runs 5-10x slower than alone):

      * glibc serves every array above 128 KB with its own mmap and returns it with munmap; each munmap interrupts
        every core the process has threads on (BLAS pool, HIP runtime, clustering workers).  Raising the mmap and
        trim thresholds keeps those arrays in the heap, where they are recycled without system calls;
      * the matrix products of one recording are tiny (1000 x 256 x 128): a 64-thread BLAS pool costs more to wake
        than it saves.

    Returns a context manager that limits the BLAS pools (a no-op without ``threadpoolctl``).  Both are process-wide:
    an application that embeds ``diarize()`` and manages its own heap / BLAS settings switches them off with
    ``VBX_AMD_TUNE_HOST=0``."""
    import contextlib
    import ctypes
    if os.environ.get('VBX_AMD_TUNE_HOST', '1') == '0':
        return contextlib.nullcontext()
    try:
        libc = ctypes.CDLL(None)
        libc.mallopt(ctypes.c_int(-3), ctypes.c_int(1 << 30))     # M_MMAP_THRESHOLD (glibc clamps it to 32 MB)
        libc.mallopt(ctypes.c_int(-1), ctypes.c_int((1 << 31) - 1))   # M_TRIM_THRESHOLD
    except (OSError, AttributeError):
        pass
    return _small_blas_pool()


def _read_recordings(ark_path):
    """[(recording, names, x[T, D])] in archive order; x-vectors of a recording are consecutive (vbhmm.py:119)."""
    return read_vec_flt_ark_grouped(ark_path)


def _rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1))


def _write_rttm_files(args, file_name, st, segs_dict):
    """vbhmm.py:166-179 for one recording."""
    assert np.all(segs_dict[file_name][0] == st['seg_names'])                 # vbhmm.py:166
    start, end = segs_dict[file_name][1].T
    starts, ends, out_labels = merge_adjacent_labels(start, end, st['labels1st'])
    os.makedirs(args.out_rttm_dir, exist_ok=True)
    with open(os.path.join(args.out_rttm_dir, f'{file_name}.rttm'), 'w') as fp:
        write_rttm(fp, file_name, out_labels, starts, ends)
    if args.output_2nd and args.init.endswith('VB') and st['labels2nd'] is not None:
        starts, ends, out_labels2 = merge_adjacent_labels(start, end, st['labels2nd'])
        second = f'{args.out_rttm_dir}2nd'
        os.makedirs(second, exist_ok=True)
        with open(os.path.join(second, f'{file_name}.rttm'), 'w') as fp:
            write_rttm(fp, file_name, out_labels2, starts, ends)


def diarize(args, stages=None, log=print):
    """The whole archive.  ``stages`` defaults to ``DeviceStages()`` (the CPU test-suite injects an object with the same
    four methods built on its checkers).  Returns ``({recording: dict(labels1st, labels2nd, n_iters, thr)}, timing)``
    for the recordings of this rank.

    A recording the device path cannot take (more than ``VBX_MAX_SPEAKERS`` = 16 384 AHC clusters, or one whose batch
    fails) does not take the archive down with it: every other recording is diarized and written, then a
    ``RuntimeError`` names the ones left out.  RTTM files are written as soon as their labels exist."""
    from .batch import shard_recordings
    from ._capi import MAX_SPEAKERS, VbxError
    assert 0 <= args.loopP <= 1, f'Expecting loopP between 0 and 1, got {args.loopP} instead.'
    t_start = time.perf_counter()
    segs_dict = read_xvector_timing_dict(args.segments_file)
    models = load_models(args.xvec_transform, args.plda_file)
    recordings = _read_recordings(args.xvec_ark_file)
    rank, world = _rank_world()
    if world > 1:                                                 # AHC is O(T^2), the VB loop O(T S): deal by T^2
        assignment = shard_recordings([float(len(r[1])) ** 2 for r in recordings], world)
        recordings = [r for k, r in enumerate(recordings) if assignment[k] == rank]
    t_read = time.perf_counter()
    want_vb = args.init.endswith('VB')
    own_stages = stages is None
    if own_stages:
        stages = DeviceStages()
    failed = {}
    try:
        # ---- stage 0: every x-vector of this rank to the device, projections for all recordings at once ---------------
        stages.project(recordings, models, args.lda_dim)
        t_proj = time.perf_counter()

        # ---- stage 1: AHC initialisation -----------------------------------------------------------------------------
        # A few recordings at a time on worker threads, each with its own device stream: the device calls run without
        # the interpreter lock and the nearest-neighbour chains of different recordings run on different CUs.
        from concurrent.futures import ThreadPoolExecutor

        def prepare(k):
            file_name, seg_names, _ = recordings[k]
            labels1st, thr = stages.ahc(k, args.threshold)
            st = dict(labels1st=labels1st, labels2nd=None, thr=thr, n_iters=0, seg_names=seg_names)
            if not want_vb:
                _write_rttm_files(args, file_name, st, segs_dict)        # AHC only: the result is final
            return file_name, st

        # the Python glue between the native calls limits the useful threads for short recordings; every thread holds a
        # T x T score matrix on the device while it works (T = 20 000: 3.2 GB)
        n_workers = int(os.environ.get('VBX_AMD_DRIVER_THREADS', '6'))
        n_workers = max(1, min(n_workers, (os.cpu_count() or 2) - 1))
        state = {}
        for rec in recordings:
            log(rec[0])                                               # vbhmm.py:121
        with tune_host_process(), ThreadPoolExecutor(max_workers=n_workers) as pool:
            for file_name, st in pool.map(prepare, range(len(recordings))):      # results in archive order
                state[file_name] = st
        if hasattr(stages, 'close_thread_contexts'):
            stages.close_thread_contexts()
        t_ahc = time.perf_counter()

        # ---- stage 2: every recording of this rank in one batch ------------------------------------------------------
        if want_vb and recordings:
            hyper = dict(loopProb=args.loopP, Fa=args.Fa, Fb=args.Fb)
            common = dict(init_smoothing=args.init_smoothing, maxIters=40, epsilon=1e-6,
                          precision=getattr(args, 'precision', 'fp64'), **hyper)
            names = [r[0] for r in recordings]
            ks = []
            for k, name in enumerate(names):
                n_clusters = int(np.max(state[name]['labels1st'])) + 1
                if n_clusters > MAX_SPEAKERS:
                    failed[name] = f'AHC left {n_clusters} clusters, the device path takes at most {MAX_SPEAKERS}'
                else:
                    ks.append(k)

            def finish(k, result):
                st = state[names[k]]
                st['labels1st'], st['labels2nd'], st['n_iters'] = result
                _write_rttm_files(args, names[k], st, segs_dict)

            try:
                for k, result in zip(ks, stages.vb(ks, [state[names[k]]['labels1st'] for k in ks], **common)):
                    finish(k, result)
            except VbxError as exc:          # one recording at a time, so that a bad one only costs itself
                log(f'batched VB-HMM failed ({exc}); retrying the recordings one by one')
                for k in ks:
                    try:
                        finish(k, stages.vb([k], [state[names[k]]['labels1st']], **common)[0])
                    except VbxError as exc_k:
                        failed[names[k]] = str(exc_k)
        t_vb = time.perf_counter()
    finally:
        if own_stages:
            stages.close()
    for name in failed:
        state.pop(name, None)
    for st in state.values():
        st.pop('seg_names', None)
    t_end = time.perf_counter()
    timing = dict(read=t_read - t_start, project=t_proj - t_read, ahc=t_ahc - t_proj, vb=t_vb - t_ahc, rttm=t_end - t_vb,
                  total=t_end - t_start, recordings=len(state), xvectors=int(sum(len(st['labels1st']) for st in state.values())),
                  rank=rank, world=world)
    if failed:
        raise RuntimeError('no RTTM written for: ' + '; '.join(f'{k} ({v})' for k, v in failed.items()))
    return state, timing


def build_parser():
    """The arguments of vbhmm.py:54-101, plus the knobs of this implementation."""
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument('--init', required=True, type=str, choices=['AHC', 'AHC+VB'],
                   help='AHC for using only AHC or AHC+VB for VB-HMM after AHC initilization')
    p.add_argument('--out-rttm-dir', required=True, type=str, help='Directory to store output rttm files')
    p.add_argument('--xvec-ark-file', required=True, type=str,
                   help='Kaldi ark file with x-vectors from one or more input recordings (all x-vectors of one '
                        'recording consecutive)')
    p.add_argument('--segments-file', required=True, type=str, help='File with x-vector timing info')
    p.add_argument('--xvec-transform', required=True, type=str, help='x-vector transformation: h5 (or npz) file')
    p.add_argument('--plda-file', required=True, type=str, help='PLDA model in Kaldi format')
    p.add_argument('--threshold', required=True, type=float, help='threshold (bias) used for AHC')
    p.add_argument('--lda-dim', required=True, type=int, help='x-vectors are reduced to this dimensionality for VB-HMM')
    p.add_argument('--Fa', required=True, type=float, help='Parameter of VB-HMM (see VBx.VBx)')
    p.add_argument('--Fb', required=True, type=float, help='Parameter of VB-HMM (see VBx.VBx)')
    p.add_argument('--loopP', required=True, type=float, help='Parameter of VB-HMM (see VBx.VBx)')
    p.add_argument('--target-energy', required=False, type=float, default=1.0,
                   help='accepted for compatibility (only used by the PLDA-scoring AHC the driver does not call)')
    p.add_argument('--init-smoothing', required=False, type=float, default=5.0,
                   help='smoothing of the hard AHC labels into the initial soft assignments')
    p.add_argument('--output-2nd', required=False, type=bool, default=False,
                   help='Output also second most likely speaker of VB-HMM')
    p.add_argument('--precision', default='fp64', choices=['fp64', 'fp32', 'fp32-split'],
                   help='arithmetic of the VB-HMM kernels (fp64 = the reference\'s dtype and iteration counts)')
    p.add_argument('--timing', action='store_true', help='print a JSON line with the stage timings of this rank')
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    dist = None
    if int(os.environ.get('WORLD_SIZE', 1)) > 1:                  # launched by torchrun: one process per GPU
        import torch
        import torch.distributed as dist
        # one GPU per rank (LOCAL_RANK); VBX_AMD_DEVICE / VBX_AMD_DIST_BACKEND override both, e.g. several ranks sharing
        # one device over gloo (RCCL refuses two ranks on one GPU)
        if torch.cuda.is_available():
            os.environ.setdefault('VBX_AMD_DEVICE', os.environ.get('LOCAL_RANK', '0'))
            torch.cuda.set_device(int(os.environ['VBX_AMD_DEVICE']))
        dist.init_process_group(os.environ.get('VBX_AMD_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo'))
    error = None
    timing = None
    try:
        try:
            _state, timing = diarize(args)
        except Exception as exc:           # the other ranks wait in the barrier below: meet them there first, then fail
            error = exc
        if dist is not None:
            dist.barrier()
    finally:
        if dist is not None and dist.is_initialized():
            dist.destroy_process_group()
    if error is not None:
        print(f'vbhmm: {type(error).__name__}: {error}', file=sys.stderr)
        return 1
    if args.timing:
        import json
        print(json.dumps(timing))
    return 0


if __name__ == '__main__':
    sys.exit(main())
