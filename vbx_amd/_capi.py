"""ctypes binding of libvbx_hip.so (include/vbx_hip.h).  No CPU fallback: if the library or
a gfx950 device is missing, every entry point raises."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from . import build as _build

VBX_F32, VBX_F64 = 0, 1
PREC_FP32, PREC_FP64 = 0, 1
FB_AUTO, FB_SEQUENTIAL, FB_CHUNKED = 0, 1, 2
OPT_FB_ALGO, OPT_CHECK_EVERY, OPT_PROFILE, OPT_CHUNK_FRAMES, OPT_FUSE, OPT_SCAN_GROUP = 1, 2, 3, 4, 5, 6
OPT_TWO_LEVEL_FROM, OPT_STREAMS, OPT_SPLIT_TILES, OPT_SCAN_GROUP2, OPT_THREE_LEVEL_FROM = 8, 10, 11, 12, 13
OPT_GEMM = 14                # how the fp32 path multiplies: GEMM_EXACT (f32 MFMA) | GEMM_SPLIT (f16 operand pairs, vbx_split.hpp)
GEMM_EXACT, GEMM_SPLIT = 0, 1
OPT_ASYNC_UPLOAD = 15        # setters only enqueue; one synchronize when the next run begins (Batch.set_async_upload)
K_NAMES = ['prep', 'mstep_acc', 'mstep_fin', 'loglik', 'fb', 'fb_aux', 'post', 'iter_fin', 'chunk_loglik',
           'chunk_post']
MAX_SPEAKERS = 16384

ABI_SYMBOLS = [
    'vbx_abi_version', 'vbx_create', 'vbx_destroy', 'vbx_last_error', 'vbx_device_info',
    'vbx_batch_create', 'vbx_batch_create_streams', 'vbx_batch_destroy', 'vbx_batch_set_option', 'vbx_batch_set_recording',
    'vbx_batch_run', 'vbx_batch_get_result', 'vbx_batch_last_run_ms', 'vbx_batch_kernel_times',
    'vbx_run', 'vbx_forward_backward', 'vbx_forward_backward_dense', 'vbx_mstep', 'vbx_loglik',
    'vbx_cos_similarity', 'vbx_scores_upload', 'vbx_scores_count', 'vbx_scores_get', 'vbx_scores_get_condensed', 'vbx_scores_linkage_average',
    'vbx_linkage_average', 'vbx_linkage_average_fastcluster', 'vbx_fcluster_distance', 'vbx_ark_index', 'vbx_gather_rows', 'vbx_batch_streams',
    'vbx_scores_two_gmm_calib',
    'vbx_scores_destroy',
    'vbx_xvectors_project', 'vbx_xvectors_get', 'vbx_xvectors_destroy', 'vbx_cos_similarity_resident',
    'vbx_batch_set_recording_resident', 'vbx_batch_get_labels', 'vbx_batch_set_recording_shared',
    'vbx_batch_gemm_in_effect',
    'vbx_batch_stream_of', 'vbx_batch_sync_uploads', 'vbx_batch_get_results', 'vbx_host_alloc', 'vbx_host_free',
]


class VbxError(RuntimeError):
    pass


_lib = None


def library_path() -> str:
    return _build.LIB


def load():
    """Load (building if the sources are newer) the HIP library.  Raises when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.path.exists(path) or (_build.is_stale() and os.environ.get('VBX_AMD_NO_REBUILD') != '1'):
        try:
            path = _build.build()
        except Exception as exc:  # a stale-but-present library is still usable on a box without hipcc
            if not os.path.exists(path):
                raise VbxError(f'libvbx_hip.so is not built and cannot be built here: {exc}') from exc
    want = os.environ.get('VBX_AMD_HW_QUEUES') or '8'       # before the HIP runtime starts (see vbx_host_state.hpp); '0': hands off
    if want != '0':
        os.environ.setdefault('GPU_MAX_HW_QUEUES', want)
    lib = C.CDLL(path)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    lib.vbx_abi_version.restype = C.c_int
    lib.vbx_create.argtypes = [C.POINTER(vp), C.c_int]
    lib.vbx_destroy.argtypes = [vp]
    lib.vbx_last_error.argtypes = [vp]
    lib.vbx_last_error.restype = C.c_char_p
    lib.vbx_device_info.argtypes = [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(i64)]
    lib.vbx_batch_create.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(i32), i32, C.c_int, C.c_int,
                                     C.POINTER(vp)]
    lib.vbx_batch_create_streams.argtypes = [vp, C.c_int, C.POINTER(i64), C.POINTER(i32), i32, C.c_int, C.c_int, C.c_int,
                                             C.POINTER(vp)]
    lib.vbx_batch_destroy.argtypes = [vp]
    lib.vbx_batch_set_option.argtypes = [vp, C.c_int, i64]
    lib.vbx_batch_set_recording.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp, vp, dbl, dbl, dbl]
    lib.vbx_batch_run.argtypes = [vp, C.c_int, dbl]
    lib.vbx_batch_get_result.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int, C.POINTER(C.c_int),
                                         C.POINTER(C.c_int), vp, vp]
    lib.vbx_batch_last_run_ms.argtypes = [vp, C.POINTER(dbl), C.POINTER(C.c_int)]
    lib.vbx_batch_kernel_times.argtypes = [vp, vp, vp]
    lib.vbx_batch_streams.argtypes = [vp]
    lib.vbx_batch_gemm_in_effect.argtypes = [vp]
    lib.vbx_run.argtypes = [vp, vp, vp]
    lib.vbx_forward_backward.argtypes = [vp, i64, i32, vp, vp, vp, dbl, C.c_int, C.c_int, vp, C.POINTER(dbl),
                                         vp, vp, vp]
    lib.vbx_forward_backward_dense.argtypes = [vp, i64, i32, vp, vp, vp, C.c_int, vp, C.POINTER(dbl), vp, vp]
    lib.vbx_mstep.argtypes = [vp, i64, i32, i32, vp, vp, vp, dbl, dbl, C.c_int, vp, vp]
    lib.vbx_loglik.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, dbl, C.c_int, vp]
    lib.vbx_cos_similarity.argtypes = [vp, i64, i32, vp, C.POINTER(vp)]
    lib.vbx_scores_upload.argtypes = [vp, i64, vp, C.POINTER(vp)]
    lib.vbx_scores_count.argtypes = [vp]
    lib.vbx_scores_get.argtypes = [vp, i64, i64, vp]
    lib.vbx_scores_get_condensed.argtypes = [vp, i64, dbl, vp]
    lib.vbx_scores_linkage_average.argtypes = [vp, i64, vp]
    lib.vbx_linkage_average.argtypes = [i64, vp, vp]
    lib.vbx_linkage_average_fastcluster.argtypes = [i64, vp, vp]
    lib.vbx_fcluster_distance.argtypes = [i64, vp, dbl, vp]
    lib.vbx_ark_index.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp]
    lib.vbx_ark_index.restype = i64
    lib.vbx_gather_rows.argtypes = [vp, i64, vp, i64, i64, vp]
    lib.vbx_scores_two_gmm_calib.argtypes = [vp, i32, C.POINTER(dbl), vp]
    lib.vbx_scores_destroy.argtypes = [vp]
    lib.vbx_xvectors_project.argtypes = [vp, i64, i32, i32, i32, vp, C.c_int, vp, vp, vp, vp, vp, C.POINTER(vp)]
    lib.vbx_xvectors_get.argtypes = [vp, C.c_int, i64, i64, vp]
    lib.vbx_xvectors_destroy.argtypes = [vp]
    lib.vbx_cos_similarity_resident.argtypes = [vp, vp, i64, i64, C.POINTER(vp)]
    lib.vbx_batch_set_recording_resident.argtypes = [vp, C.c_int, vp, i64, vp, dbl, vp, dbl, dbl, dbl]
    lib.vbx_batch_get_labels.argtypes = [vp, C.c_int, vp, vp]
    lib.vbx_batch_sync_uploads.argtypes = [vp]
    lib.vbx_batch_stream_of.argtypes = [vp, C.c_int]
    lib.vbx_batch_get_results.argtypes = [vp, C.c_int, vp]
    lib.vbx_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.vbx_host_free.argtypes = [vp]
    lib.vbx_batch_set_recording_shared.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, vp, dbl, dbl, dbl]
    for name in ABI_SYMBOLS:
        fn = getattr(lib, name)          # AttributeError here = the .so does not export the ABI
        if name in ('vbx_scores_count', 'vbx_ark_index'):
            fn.restype = C.c_int64
        elif name not in ('vbx_last_error', 'vbx_abi_version'):
            fn.restype = C.c_int
    _lib = lib
    return lib


class Fetch(C.Structure):
    """vbx_fetch (include/vbx_hip.h): one recording's result pointers for vbx_batch_get_results."""
    _fields_ = [('rec', C.c_int32), ('gamma', C.c_void_p), ('pi', C.c_void_p), ('Li', C.c_void_p), ('li_cap', C.c_int32),
                ('n_iters', C.c_int32), ('warned', C.c_int32), ('alpha', C.c_void_p), ('invL', C.c_void_p)]


class _PinnedBlock:
    """A block of pinned host memory from the library's pool (vbx_host_alloc); goes back to the pool with the last array on it."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        rc = load().vbx_host_alloc(int(nbytes), C.byref(p))
        if rc != 0:
            raise VbxError(f'vbx_host_alloc({nbytes}) failed ({rc}): ' + (load().vbx_last_error(None) or b'').decode())
        self.ptr, self.nbytes = p.value, int(nbytes)

    def __del__(self):
        if sys.is_finalizing() or not getattr(self, 'ptr', None):
            return
        try:
            load().vbx_host_free(self.ptr)
        except Exception:
            pass
        self.ptr = None


def pinned_arrays(shapes, dtype=np.float64):
    """float64 arrays of the given shapes on ONE block of pinned host memory (the destination of choice for results: a
    copy out of HBM into such an array is a plain DMA; into ordinary memory it is staged and three to ten times slower).
    The block returns to the library's pool when the last of the arrays (or a view of one) is gone."""
    item = np.dtype(dtype).itemsize
    sizes = [int(np.prod(s)) * item for s in shapes]
    offs = np.concatenate([[0], np.cumsum([(n + 63) // 64 * 64 for n in sizes])])
    block = _PinnedBlock(max(int(offs[-1]), 64))
    raw = (C.c_char * block.nbytes).from_address(block.ptr)
    raw._vbx_owner = block                              # np.frombuffer keeps `raw` alive, `raw` keeps the block
    out = []
    for s, n, o in zip(shapes, sizes, offs):
        out.append(np.frombuffer(raw, dtype=dtype, count=n // item, offset=int(o)).reshape(s))
    return out


def experiment_env(name):
    """An environment variable that exists for A/B measurements only (second table): read only when
    VBX_AMD_EXPERIMENT=1, so that a stray variable in a production environment cannot change what the library runs."""
    return os.environ.get(name) if os.environ.get('VBX_AMD_EXPERIMENT') == '1' else None


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


SPLIT_NAMES = ('fp32-split', 'fp32s', 'f32-split')       # fp32 storage, the two GEMMs on f16 operand pairs (VBX_OPT_GEMM)


def precision_code(precision) -> int:
    if precision in (PREC_FP32, 'fp32', 'f32', 'float32', np.float32) or precision in SPLIT_NAMES:
        return PREC_FP32
    if precision in (PREC_FP64, 'fp64', 'f64', 'float64', np.float64):
        return PREC_FP64
    raise ValueError(f'unknown precision {precision!r}')


class Context:
    """One device + one HIP stream (vbx_ctx)."""

    def __init__(self, device: int = 0):
        self._lib = load()
        h = C.c_void_p()
        rc = self._lib.vbx_create(C.byref(h), int(device))
        if rc != 0:
            raise VbxError(f'vbx_create(device={device}) failed ({rc}): '
                           f'{self._lib.vbx_last_error(None).decode()}')
        self._h = h
        self.device = int(device)

    def check(self, rc: int, what: str):
        if rc != 0:
            raise VbxError(f'{what} failed ({rc}): {self._lib.vbx_last_error(self._h).decode()}')

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, hbm = C.c_int(), C.c_int64()
        self.check(self._lib.vbx_device_info(self._h, name, 256, C.byref(cus), C.byref(hbm)), 'vbx_device_info')
        return {'name': name.value.decode(), 'compute_units': cus.value, 'hbm_bytes': hbm.value}

    def close(self):
        if getattr(self, '_h', None):
            self._lib.vbx_destroy(self._h)
            self._h = None

    def __del__(self):
        if sys.is_finalizing():          # the HIP runtime may already be gone: leave the device memory to the process exit
            return
        try:
            self.close()
        except Exception:
            pass

    # ---- step-level entry points (parity tests) -------------------------------------------
    def forward_backward(self, lls, pi, loopProb, ip=None, precision='fp64', fb_algo=FB_AUTO, want_logs=False):
        lls = _f64(lls)
        T, S = lls.shape
        pi = _f64(pi)
        ip = _f64(ip)
        gamma = np.empty((T, S))
        entered = np.empty(S)
        tll = C.c_double()
        lfw = np.empty((T, S)) if want_logs else None
        lbw = np.empty((T, S)) if want_logs else None
        self.check(self._lib.vbx_forward_backward(self._h, T, S, _ptr(lls), _ptr(pi), _ptr(ip), float(loopProb),
                                                  precision_code(precision), int(fb_algo), _ptr(gamma),
                                                  C.byref(tll), _ptr(entered), _ptr(lfw), _ptr(lbw)),
                   'vbx_forward_backward')
        return gamma, tll.value, entered, lfw, lbw

    def forward_backward_dense(self, lls, tr, ip, precision='fp64', want_logs=True):
        """forward_backward (VBx.py:146-175) for any S x S transition matrix."""
        lls, tr, ip = _f64(lls), _f64(tr), _f64(ip)
        T, S = lls.shape
        assert tr.shape == (S, S) and ip.shape == (S,)
        gamma = np.empty((T, S))
        lfw = np.empty((T, S)) if want_logs else None
        lbw = np.empty((T, S)) if want_logs else None
        tll = C.c_double()
        self.check(self._lib.vbx_forward_backward_dense(self._h, T, S, _ptr(lls), _ptr(tr), _ptr(ip),
                                                        precision_code(precision), _ptr(gamma), C.byref(tll),
                                                        _ptr(lfw), _ptr(lbw)), 'vbx_forward_backward_dense')
        return gamma, tll.value, lfw, lbw

    def mstep(self, X, Phi, gamma, Fa, Fb, precision='fp64'):
        X, Phi, gamma = _f64(X), _f64(Phi), _f64(gamma)
        T, D = X.shape
        S = gamma.shape[1]
        alpha, invL = np.empty((S, D)), np.empty((S, D))
        self.check(self._lib.vbx_mstep(self._h, T, S, D, _ptr(X), _ptr(Phi), _ptr(gamma), float(Fa), float(Fb),
                                       precision_code(precision), _ptr(alpha), _ptr(invL)), 'vbx_mstep')
        return alpha, invL

    def loglik(self, X, Phi, alpha, invL, Fa, precision='fp64'):
        X, Phi, alpha, invL = _f64(X), _f64(Phi), _f64(alpha), _f64(invL)
        T, D = X.shape
        S = alpha.shape[0]
        out = np.empty((T, S))
        self.check(self._lib.vbx_loglik(self._h, T, S, D, _ptr(X), _ptr(Phi), _ptr(alpha), _ptr(invL), float(Fa),
                                        precision_code(precision), _ptr(out)), 'vbx_loglik')
        return out


class XVectors:
    """The x-vectors of an archive after the driver's projections, resident in HBM (vbx_xvectors): ``xproj`` (the rows
    cos_similarity works on, vbhmm.py:125-129) and ``fea`` (the input of VBx(), vbhmm.py:153)."""

    def __init__(self, ctx: Context, x, mean1, lda, mean2, plda_mu, plda_tr, fea_dim):
        self.ctx, self._lib = ctx, ctx._lib
        x = np.ascontiguousarray(x)
        if x.dtype != np.float32:
            x = np.ascontiguousarray(x, dtype=np.float64)
        mean1, lda, mean2, plda_mu, plda_tr = _f64(mean1), _f64(lda), _f64(mean2), _f64(plda_mu), _f64(plda_tr)
        n, din = x.shape
        dl = lda.shape[1]
        assert lda.shape == (din, dl) and mean1.shape == (din,) and mean2.shape == (dl,) and plda_mu.shape == (dl,)
        assert plda_tr.shape == (dl, dl) and 0 < fea_dim <= dl
        h = C.c_void_p()
        ctx.check(self._lib.vbx_xvectors_project(ctx._h, n, din, dl, int(fea_dim), _ptr(x),
                                                 VBX_F32 if x.dtype == np.float32 else VBX_F64, _ptr(mean1), _ptr(lda),
                                                 _ptr(mean2), _ptr(plda_mu), _ptr(plda_tr), C.byref(h)),
                  'vbx_xvectors_project')
        self._h, self.n, self.dl, self.fea_dim = h, n, dl, int(fea_dim)

    def get(self, which, row0=0, nrows=None):
        """rows of 'xproj' or 'fea' as a float64 array (tests; the driver never needs them on the host)."""
        nrows = self.n - row0 if nrows is None else nrows
        out = np.empty((nrows, self.dl if which == 'xproj' else self.fea_dim))
        self.ctx.check(self._lib.vbx_xvectors_get(self._h, 0 if which == 'xproj' else 1, int(row0), int(nrows), _ptr(out)),
                       'vbx_xvectors_get')
        return out

    def close(self):
        if getattr(self, '_h', None):
            self._lib.vbx_xvectors_destroy(self._h)
            self._h = None

    def __del__(self):
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Scores:
    """A vector of float64 scores resident in HBM (vbx_scores): the T x T cosine similarities of the AHC
    initialisation, or any scores handed to the two-Gaussian calibration."""

    def __init__(self, ctx: Context, handle):
        self.ctx, self._lib, self._h = ctx, ctx._lib, handle

    @classmethod
    def cos_similarity(cls, ctx: Context, x):
        x = _f64(x)
        h = C.c_void_p()
        ctx.check(ctx._lib.vbx_cos_similarity(ctx._h, x.shape[0], x.shape[1], _ptr(x), C.byref(h)), 'vbx_cos_similarity')
        return cls(ctx, h)

    @classmethod
    def cos_similarity_resident(cls, ctx: Context, xv: 'XVectors', row0, T):
        """cos_similarity of rows [row0, row0 + T) of the projected x-vectors already in HBM."""
        h = C.c_void_p()
        ctx.check(ctx._lib.vbx_cos_similarity_resident(ctx._h, xv._h, int(row0), int(T), C.byref(h)),
                  'vbx_cos_similarity_resident')
        return cls(ctx, h)

    @classmethod
    def upload(cls, ctx: Context, s):
        s = _f64(s).reshape(-1)
        h = C.c_void_p()
        ctx.check(ctx._lib.vbx_scores_upload(ctx._h, s.size, _ptr(s), C.byref(h)), 'vbx_scores_upload')
        return cls(ctx, h)

    def __len__(self):
        return int(self._lib.vbx_scores_count(self._h))

    def get(self, offset=0, count=None, out=None):
        count = len(self) - offset if count is None else count
        if out is None:
            out = np.empty(count)
        assert out.dtype == np.float64 and out.flags.c_contiguous and out.size == count
        self.ctx.check(self._lib.vbx_scores_get(self._h, int(offset), int(count), _ptr(out)), 'vbx_scores_get')
        return out

    def get_condensed(self, T, scale=1.0):
        """Strict upper triangle of ``scale * S`` (S = these scores as a T x T matrix), row by row: the vector form
        of ``scipy.spatial.distance.squareform``."""
        out = np.empty(int(T) * (int(T) - 1) // 2)
        self.ctx.check(self._lib.vbx_scores_get_condensed(self._h, int(T), float(scale), _ptr(out)),
                       'vbx_scores_get_condensed')
        return out

    def linkage_average(self, T):
        """``linkage(squareform(-S), 'average')`` for the T x T similarities held here, computed on the device (the scores
        are consumed); the linkage matrix [T - 1][4], bit for bit what ``linkage_average`` gives on the host."""
        Z = np.empty((max(int(T) - 1, 0), 4))
        self.ctx.check(self._lib.vbx_scores_linkage_average(self._h, int(T), _ptr(Z)), 'vbx_scores_linkage_average')
        return Z

    def two_gmm_calib(self, niters=20, want_llr=True):
        thr = C.c_double()
        llr = np.empty(len(self)) if want_llr else None
        self.ctx.check(self._lib.vbx_scores_two_gmm_calib(self._h, int(niters), C.byref(thr), _ptr(llr)),
                       'vbx_scores_two_gmm_calib')
        return thr.value, llr

    def close(self):
        if getattr(self, '_h', None):
            self._lib.vbx_scores_destroy(self._h)
            self._h = None

    def __del__(self):
        if sys.is_finalizing():          # the HIP runtime may already be gone: leave the device memory to the process exit
            return
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """A set of recordings resident in HBM (vbx_batch)."""

    def __init__(self, ctx: Context, T, S, D: int, precision='fp32', max_iters: int = 40, streams: int = 0):
        """``streams``: HIP streams (sub-batches) of the batch, 0 = the library's choice (vbx_batch_create_streams); a sweep
        over one long recording -- few recordings, many chunks -- asks for its streams here."""
        self.ctx = ctx
        self._lib = ctx._lib
        self.T = [int(t) for t in T]
        self.S = [int(s) for s in S]
        self.D = int(D)
        self.n = len(self.T)
        self.precision = precision_code(precision)
        self.max_iters = int(max_iters)
        h = C.c_void_p()
        Ta = (C.c_int64 * self.n)(*self.T)
        Sa = (C.c_int32 * self.n)(*self.S)
        ctx.check(self._lib.vbx_batch_create_streams(ctx._h, self.n, Ta, Sa, self.D, self.precision, self.max_iters,
                                                     int(streams or 0), C.byref(h)), 'vbx_batch_create_streams')
        self._h = h
        self._held = []                                    # arrays an asynchronous upload still reads (set_async_upload)
        self._async = False
        def choice(var, value, table):
            if value not in table:
                raise ValueError(f'{var}={value!r}: expected one of {", ".join(map(repr, table))}')
            return table[value]
        algo = experiment_env('VBX_AMD_FB_ALGO')          # 'sequential' | 'chunked' (default: auto)
        if algo:
            self.set_option(OPT_FB_ALGO, choice('VBX_AMD_FB_ALGO', algo, {'auto': FB_AUTO, 'sequential': FB_SEQUENTIAL,
                                                                          'chunked': FB_CHUNKED}))
        if experiment_env('VBX_AMD_TWO_LEVEL_FROM') is not None:
            self.set_option(OPT_TWO_LEVEL_FROM, int(experiment_env('VBX_AMD_TWO_LEVEL_FROM')))
        group = experiment_env('VBX_AMD_SCAN_GROUP')      # chunks per group of the two-level boundary walk
        if group is not None:
            self.set_option(OPT_SCAN_GROUP, int(group))
        group2 = experiment_env('VBX_AMD_SCAN_GROUP2')    # level-2 groups of the three-level walk (0 auto, 1 off)
        if group2 is not None:
            self.set_option(OPT_SCAN_GROUP2, int(group2))
        if experiment_env('VBX_AMD_THREE_LEVEL_FROM') is not None:
            self.set_option(OPT_THREE_LEVEL_FROM, int(experiment_env('VBX_AMD_THREE_LEVEL_FROM')))
        split = experiment_env('VBX_AMD_SPLIT_TILES')     # 1 / 2: half-tile re-runs on / off (0: the library's choice)
        if split is not None:
            self.set_option(OPT_SPLIT_TILES, int(split))
        # VBX_OPT_GEMM: the precision argument decides where it names a mode ('fp32-split'); VBX_AMD_GEMM = exact | split
        # decides for a plain 'fp32' -- an explicit argument is never overridden by the environment, in either direction
        gemm = os.environ.get('VBX_AMD_GEMM')
        if gemm:
            choice('VBX_AMD_GEMM', gemm, {'exact': GEMM_EXACT, 'split': GEMM_SPLIT})
        if isinstance(precision, str) and precision in SPLIT_NAMES:
            gemm = 'split'
        if gemm:
            self.set_option(OPT_GEMM, {'exact': GEMM_EXACT, 'split': GEMM_SPLIT}[gemm])
        fuse = experiment_env('VBX_AMD_FUSE')             # '0' keeps every stage in its own kernel
        if fuse is not None:
            self.set_option(OPT_FUSE, int(fuse))

    def profile_kernels(self, names=None):
        """HIP events around every launch (names=None), around the given kernel classes only, or off ([])."""
        if names is None:
            self.set_option(OPT_PROFILE, 1)
        else:
            self.set_option(OPT_PROFILE, 2 * sum(1 << K_NAMES.index(n) for n in names))

    def set_option(self, option: int, value: int):
        self.ctx.check(self._lib.vbx_batch_set_option(self._h, int(option), int(value)), 'vbx_batch_set_option')

    def set_async_upload(self, on=True):
        """Uploads without a synchronize per recording (VBX_OPT_ASYNC_UPLOAD): set_recording only enqueues; the arrays it was
        given are kept alive here until the next run() / sync_uploads()."""
        self.set_option(OPT_ASYNC_UPLOAD, 1 if on else 0)
        self._async = bool(on)
        if not on:
            self._held.clear()

    def stream_of(self, b) -> int:
        """The stream (sub-batch) recording b was dealt to: recordings of different streams may be set from different threads."""
        return int(self._lib.vbx_batch_stream_of(self._h, int(b)))

    def sync_uploads(self):
        self.ctx.check(self._lib.vbx_batch_sync_uploads(self._h), 'vbx_batch_sync_uploads')
        self._held.clear()

    def set_recording(self, b, X, Phi, pi0, gamma0, loopProb, Fa, Fb, alpha0=None, invL0=None):
        X = np.ascontiguousarray(X)
        if X.dtype != np.float32:
            X = np.ascontiguousarray(X, dtype=np.float64)
        gamma0 = np.ascontiguousarray(gamma0)
        if gamma0.dtype != np.float32:
            gamma0 = np.ascontiguousarray(gamma0, dtype=np.float64)
        assert X.shape == (self.T[b], self.D) and gamma0.shape == (self.T[b], self.S[b])
        Phi, pi0, alpha0, invL0 = _f64(Phi), _f64(pi0), _f64(alpha0), _f64(invL0)
        assert Phi.shape == (self.D,) and pi0.shape == (self.S[b],)
        self.ctx.check(self._lib.vbx_batch_set_recording(
            self._h, int(b), _ptr(X), VBX_F32 if X.dtype == np.float32 else VBX_F64, _ptr(Phi), _ptr(pi0),
            _ptr(gamma0), VBX_F32 if gamma0.dtype == np.float32 else VBX_F64, _ptr(alpha0), _ptr(invL0),
            float(loopProb), float(Fa), float(Fb)), 'vbx_batch_set_recording')
        if self._async:
            self._held.append((X, gamma0))

    def set_recording_shared(self, b, src, pi0, gamma0, loopProb, Fa, Fb, alpha0=None, invL0=None):
        """Recording b on the x-vectors (rho, Phi) of recording ``src`` of this batch, set before: one point of an
        Fa / Fb / loopProb sweep over one recording (vbx_batch_set_recording_shared)."""
        gamma0 = np.ascontiguousarray(gamma0)
        if gamma0.dtype != np.float32:
            gamma0 = np.ascontiguousarray(gamma0, dtype=np.float64)
        assert gamma0.shape == (self.T[b], self.S[b]) and self.T[b] == self.T[src]
        pi0, alpha0, invL0 = _f64(pi0), _f64(alpha0), _f64(invL0)
        assert pi0.shape == (self.S[b],)
        self.ctx.check(self._lib.vbx_batch_set_recording_shared(
            self._h, int(b), int(src), _ptr(pi0), _ptr(gamma0), VBX_F32 if gamma0.dtype == np.float32 else VBX_F64,
            _ptr(alpha0), _ptr(invL0), float(loopProb), float(Fa), float(Fb)), 'vbx_batch_set_recording_shared')
        if self._async:
            self._held.append((gamma0,))

    def set_recording_resident(self, b, xv: 'XVectors', row0, labels, init_smoothing, Phi, loopProb, Fa, Fb):
        """Recording b from resident rows of ``xv.fea`` and the AHC labels: initial responsibilities
        softmax(init_smoothing * onehot(labels)) built on the device, uniform priors (vbhmm.py:150-158)."""
        labels = np.ascontiguousarray(labels, dtype=np.int32)
        Phi = _f64(Phi)
        assert labels.shape == (self.T[b],) and Phi.shape == (self.D,)
        self.ctx.check(self._lib.vbx_batch_set_recording_resident(
            self._h, int(b), xv._h, int(row0), _ptr(labels), float(init_smoothing), _ptr(Phi), float(loopProb),
            float(Fa), float(Fb)), 'vbx_batch_set_recording_resident')

    def labels(self, b):
        """(first, second) speaker of every frame of recording b: np.argsort(-gamma, axis=1)[:, 0 / 1] computed on the
        device (vbhmm.py:160-162); second is None for a single speaker."""
        first = np.empty(self.T[b], dtype=np.int32)
        second = np.empty(self.T[b], dtype=np.int32)
        self.ctx.check(self._lib.vbx_batch_get_labels(self._h, int(b), _ptr(first), _ptr(second)), 'vbx_batch_get_labels')
        return first.astype(np.int64), (second.astype(np.int64) if self.S[b] > 1 else None)

    def n_iters(self, b):
        n, w = C.c_int(), C.c_int()
        self.ctx.check(self._lib.vbx_batch_get_result(self._h, int(b), None, None, None, 0, C.byref(n), C.byref(w), None, None),
                       'vbx_batch_get_result')
        return n.value

    def run(self, iters: int, epsilon: float = -np.inf):
        eps = float(epsilon)
        if not np.isfinite(eps):
            eps = -1e300 if eps < 0 else 1e300
        try:
            self.ctx.check(self._lib.vbx_batch_run(self._h, int(iters), eps), 'vbx_batch_run')
        finally:
            self._held.clear()                             # (the run begins by waiting for the uploads)

    def last_run_ms(self):
        ms, it = C.c_double(), C.c_int()
        self.ctx.check(self._lib.vbx_batch_last_run_ms(self._h, C.byref(ms), C.byref(it)), 'vbx_batch_last_run_ms')
        return ms.value, it.value

    @property
    def streams(self) -> int:
        """HIP streams (sub-batches) this batch runs on (VBX_OPT_STREAMS in effect)."""
        return int(self._lib.vbx_batch_streams(self._h))

    @property
    def gemm(self) -> str:
        """'split' when the iterations of the last run multiplied with f16 operand pairs (VBX_OPT_GEMM in effect), else
        'exact'."""
        return 'split' if int(self._lib.vbx_batch_gemm_in_effect(self._h)) == GEMM_SPLIT else 'exact'

    def kernel_times(self):
        ms = np.zeros(len(K_NAMES))
        n = np.zeros(len(K_NAMES), dtype=np.int64)
        self.ctx.check(self._lib.vbx_batch_kernel_times(self._h, _ptr(ms), _ptr(n)), 'vbx_batch_kernel_times')
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(K_NAMES)}

    def result(self, b, want_gamma=True, want_model=True):
        T, S, D = self.T[b], self.S[b], self.D
        gamma = np.empty((T, S)) if want_gamma else None
        pi = np.empty(S)
        Li = np.zeros(max(self.max_iters, 1))
        alpha = np.empty((S, D)) if want_model else None
        invL = np.empty((S, D)) if want_model else None
        n_iters, warned = C.c_int(), C.c_int()
        self.ctx.check(self._lib.vbx_batch_get_result(self._h, int(b), _ptr(gamma), _ptr(pi), _ptr(Li), len(Li),
                                                      C.byref(n_iters), C.byref(warned), _ptr(alpha), _ptr(invL)),
                       'vbx_batch_get_result')
        n = min(n_iters.value, self.max_iters)
        return {'gamma': gamma, 'pi': pi, 'Li': Li[:n].copy(), 'n_iters': n_iters.value,
                'warned': bool(warned.value), 'alpha': alpha, 'invL': invL}

    def results(self, recs=None, want_gamma=True, want_model=True, pinned=True):
        """``[result(b) for b in recs]`` (default: every recording) with ONE synchronize per stream (vbx_batch_get_results) and,
        with ``pinned``, the arrays of all recordings on one block of pinned host memory: the responsibilities leave HBM as
        plain DMA instead of through the runtime's staging buffers (64 recordings of T = 10 000: 25-37 ms -> a few)."""
        recs = list(range(self.n)) if recs is None else [int(b) for b in recs]
        # gamma on a pinned block of its own per recording (a caller that keeps one recording's responsibilities keeps that block,
        # not the batch's); the small arrays of all recordings share one block and are handed out as ordinary copies
        small, per = [], 4 if want_model else 2
        for b in recs:
            S, D = self.S[b], self.D
            small += [(S,), (max(self.max_iters, 1),)] + ([(S, D), (S, D)] if want_model else [])
        small = pinned_arrays(small) if pinned else [np.empty(s) for s in small]
        arrays = []
        for k, b in enumerate(recs):
            T, S = self.T[b], self.S[b]
            g = (pinned_arrays([(T, S)])[0] if pinned else np.empty((T, S))) if want_gamma else None
            arrays.append([g] + small[per * k: per * k + per])
        items = (Fetch * len(recs))()
        for k, b in enumerate(recs):
            g, p, L = arrays[k][:3]
            L[:] = 0.0
            items[k].rec = b
            items[k].gamma = g.ctypes.data if want_gamma else None
            items[k].pi = p.ctypes.data
            items[k].Li = L.ctypes.data
            items[k].li_cap = len(L)
            items[k].alpha = arrays[k][3].ctypes.data if want_model else None
            items[k].invL = arrays[k][4].ctypes.data if want_model else None
        self.ctx.check(self._lib.vbx_batch_get_results(self._h, len(recs), items), 'vbx_batch_get_results')
        out = []
        for k, b in enumerate(recs):
            g, p, L = arrays[k][:3]
            n = min(items[k].n_iters, self.max_iters)
            out.append({'gamma': g, 'pi': p.copy(), 'Li': L[:n].copy(), 'n_iters': int(items[k].n_iters),
                        'warned': bool(items[k].warned), 'alpha': arrays[k][3].copy() if want_model else None,
                        'invL': arrays[k][4].copy() if want_model else None})
        return out

    def close(self):
        if getattr(self, '_h', None):
            self._lib.vbx_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        if sys.is_finalizing():          # the HIP runtime may already be gone: leave the device memory to the process exit
            return
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def linkage_variant():
    """'scipy' (default) or 'fastcluster' ($VBX_AMD_LINKAGE): which package's arithmetic the average linkage follows."""
    v = os.environ.get('VBX_AMD_LINKAGE', 'scipy').lower()
    if v not in ('scipy', 'fastcluster'):
        raise ValueError(f"VBX_AMD_LINKAGE={v!r}: expected 'scipy' or 'fastcluster'")
    return v


def linkage_average(condensed, variant=None):
    """``linkage(condensed, method='average')`` as native host code: the (n - 1) x 4 linkage matrix.  ``variant='scipy'``
    (default, or $VBX_AMD_LINKAGE): bit for bit ``scipy.cluster.hierarchy.linkage``; ``'fastcluster'``: the form
    ``fastcluster.linkage`` computes (vbhmm.py:140-141) -- weights divided before the update, its own chain bookkeeping --
    restated from the package's published source.  Same tree, distances equal to rounding, wherever no two candidate
    distances tie.  The call releases the GIL: one recording per thread."""
    variant = variant or linkage_variant()
    if variant not in ('scipy', 'fastcluster'):
        raise ValueError(f"linkage_average: variant {variant!r}: expected 'scipy' or 'fastcluster'")
    y = np.ascontiguousarray(condensed, dtype=np.float64)
    if y.ndim != 1:
        raise ValueError('linkage_average expects a condensed distance vector')
    n = int(round((1.0 + np.sqrt(1.0 + 8.0 * y.size)) / 2.0))
    if n * (n - 1) // 2 != y.size or n < 2:
        raise ValueError(f'{y.size} is not the length n (n - 1) / 2 of a condensed distance vector')
    Z = np.empty((n - 1, 4))
    fn = load().vbx_linkage_average_fastcluster if variant == 'fastcluster' else load().vbx_linkage_average
    rc = fn(n, _ptr(y), _ptr(Z))
    if rc != 0:
        raise VbxError(f'vbx_linkage_average failed ({rc})')
    return Z


def ark_index(buf):
    """Entries of a binary Kaldi vector archive held in ``buf`` (bytes-like): arrays (key offset, key length, data
    offset, dimension, element size), or ``None`` when the archive is not of that form (text archives)."""
    raw = np.frombuffer(buf, dtype=np.uint8)
    cap = max(16, raw.size // 64)
    while True:
        ko, kl = np.empty(cap, np.int64), np.empty(cap, np.int32)
        do, dm, es = np.empty(cap, np.int64), np.empty(cap, np.int32), np.empty(cap, np.int32)
        n = load().vbx_ark_index(_ptr(raw), raw.size, cap, _ptr(ko), _ptr(kl), _ptr(do), _ptr(dm), _ptr(es))
        if n == -2:
            cap *= 4
            continue
        if n < 0:
            return None
        return ko[:n], kl[:n], do[:n], dm[:n], es[:n]


def gather_rows(raw, offsets, row_bytes, dtype):
    """``len(offsets)`` rows of ``row_bytes`` bytes of the uint8 array ``raw`` packed into a new 2-D array of ``dtype``."""
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    out = np.empty((offsets.size, row_bytes // np.dtype(dtype).itemsize), dtype=dtype)
    rc = load().vbx_gather_rows(_ptr(raw), raw.size, _ptr(offsets), offsets.size, int(row_bytes), _ptr(out))
    if rc != 0:
        raise VbxError(f'vbx_gather_rows failed ({rc}): offsets outside the buffer')
    return out


def fcluster_distance(Z, t):
    """``scipy.cluster.hierarchy.fcluster(Z, t, criterion='distance')`` as native host code (labels from 1, int32)."""
    Z = np.ascontiguousarray(Z, dtype=np.float64)
    if Z.ndim != 2 or Z.shape[1] != 4:
        raise ValueError('fcluster_distance expects an (n - 1) x 4 linkage matrix')
    n = Z.shape[0] + 1
    labels = np.empty(n, dtype=np.int32)
    rc = load().vbx_fcluster_distance(n, _ptr(Z), float(t), _ptr(labels))
    if rc != 0:
        raise VbxError(f'vbx_fcluster_distance failed ({rc}): not a linkage matrix')
    return labels


def default_context(device: int | None = None, slot: int = 0) -> Context:
    """The process-wide context of a device; ``slot`` > 0: a further one of the same device (its own streams and device
    arena: vbx_amd.batch pipelines a large batch call over two of them)."""
    if device is None:
        device = int(os.environ.get('VBX_AMD_DEVICE', os.environ.get('LOCAL_RANK', '0')))
    key = device if slot == 0 else (device, slot)
    if key not in _default_ctx:
        _default_ctx[key] = Context(device)
    return _default_ctx[key]
