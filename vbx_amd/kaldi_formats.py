"""On-disk formats either side of the VB-HMM path, without third-party
readers: the reference's driver needs ``kaldi_io`` and ``h5py`` for them (vbhmm.py:33-35, kaldi_utils.py:21-22).

    x-vector archive   Kaldi vector ark, binary ('FV ' / 'DV ') or text          vbhmm.py:117  (kaldi_io.read_vec_flt_ark)
    PLDA model         Kaldi '<Plda>' object, binary or text                       kaldi_utils.py:25-54  (read_plda)
    segments           'xvector-name recording start end' per line                 diarization_lib.py:96-113
    x-vector transform HDF5 with datasets mean1, mean2, lda (or an .npz with them) vbhmm.py:125-129
    RTTM               'SPEAKER <rec> 1 <start> <dur> <NA> <NA> <label> <NA> <NA>' vbhmm.py:48-51

Writers exist for the formats the tests and tools need to produce (ark, PLDA, segments, RTTM).
"""
from __future__ import annotations

import os
import struct

import numpy as np

__all__ = ['read_vec_flt_ark', 'read_vec_flt_ark_grouped', 'write_vec_flt_ark', 'read_plda', 'write_plda', 'read_xvector_timing_dict',
           'write_segments', 'read_xvec_transform', 'write_rttm', 'read_rttm']


# ---- Kaldi vector archives ---------------------------------------------------------------------------------
def _read_token(fd) -> bytes:
    """Key of the next archive entry: bytes up to the next space ('' at end of file)."""
    key = bytearray()
    while True:
        ch = fd.read(1)
        if ch == b'' or ch == b' ':
            return bytes(key)
        if ch in b'\r\n\t' and not key:
            continue
        key += ch


def _read_binary_vector(fd) -> np.ndarray:
    kind = fd.read(3)
    if kind == b'FV ':
        dtype = '<f4'
    elif kind == b'DV ':
        dtype = '<f8'
    else:
        raise ValueError(f'unsupported Kaldi vector type {kind!r} (expected "FV " or "DV ")')
    if fd.read(1) != b'\x04':
        raise ValueError('corrupt Kaldi vector header (size of the dimension field is not 4)')
    n = struct.unpack('<i', fd.read(4))[0]
    buf = fd.read(n * np.dtype(dtype).itemsize)
    if len(buf) != n * np.dtype(dtype).itemsize:
        raise ValueError('truncated Kaldi vector')
    return np.frombuffer(buf, dtype=dtype)


def _read_text_vector(fd) -> np.ndarray:
    """' [ v v v ]' up to and including the closing bracket."""
    buf = bytearray()
    while True:
        ch = fd.read(1)
        if ch == b'':
            raise ValueError('truncated text vector')
        if ch == b']':
            break
        buf += ch
    return np.array(buf.decode().replace('[', ' ').split(), dtype=np.float32)


def read_vec_flt_ark(path):
    """Generator of ``(key, vector)`` over a Kaldi vector archive, in file order (``kaldi_io.read_vec_flt_ark``
    as vbhmm.py:117 uses it).  Binary entries give float32 / float64 views, text entries float32."""
    with open(path, 'rb') as fd:
        while True:
            key = _read_token(fd)
            if not key:
                return
            mark = fd.read(2)
            if mark == b'\x00B':
                vec = _read_binary_vector(fd)
            else:
                fd.seek(-len(mark), os.SEEK_CUR)
                vec = _read_text_vector(fd)
            yield key.decode(), vec


def read_vec_flt_ark_grouped(path, group_key=lambda key: key.rsplit('_', 1)[0]):
    """``[(group, keys, matrix)]``: the archive's vectors stacked per group of consecutive entries with the same
    ``group_key`` -- what vbhmm.py:119-123 builds with itertools.groupby + zip + np.array, recording by recording.
    A binary archive is indexed by the library's native scanner and gathered without a Python loop over vectors;
    anything else (text archives, ragged groups) goes through ``read_vec_flt_ark``."""
    from . import _capi
    with open(path, 'rb') as fd:
        raw = fd.read()
    idx = _capi.ark_index(raw)
    if idx is None:
        import itertools
        out = []
        for name, segs in itertools.groupby(read_vec_flt_ark(path), lambda e: group_key(e[0])):
            keys, vecs = zip(*segs)
            out.append((name, np.array(keys), np.array(vecs)))
        return out
    key_off, key_len, data_off, dim, esize = idx
    keys = [raw[o:o + n].decode() for o, n in zip(key_off.tolist(), key_len.tolist())]
    groups = [group_key(k) for k in keys]
    buf = np.frombuffer(raw, dtype=np.uint8)
    out, lo = [], 0
    for hi in range(1, len(keys) + 1):
        if hi == len(keys) or groups[hi] != groups[lo]:
            d, e = int(dim[lo]), int(esize[lo])
            if np.any(dim[lo:hi] != d) or np.any(esize[lo:hi] != e):
                raise ValueError(f'x-vectors of {groups[lo]} differ in dimension or type')
            mat = _capi.gather_rows(buf, data_off[lo:hi], d * e, '<f4' if e == 4 else '<f8')
            out.append((groups[lo], np.array(keys[lo:hi]), mat))
            lo = hi
    return out


def write_vec_flt_ark(path, items, dtype=np.float32):
    """Binary Kaldi vector archive from an iterable of ``(key, vector)``."""
    tag = b'FV ' if np.dtype(dtype) == np.float32 else b'DV '
    with open(path, 'wb') as fd:
        for key, vec in items:
            v = np.ascontiguousarray(vec, dtype=np.dtype(dtype).newbyteorder('<'))
            fd.write(key.encode() + b' \x00B' + tag + b'\x04' + struct.pack('<i', v.shape[0]) + v.tobytes())


# ---- Kaldi PLDA -----------------------------------------------------------------------------------------------
def _read_binary_matrix(fd) -> np.ndarray:
    kind = fd.read(3)
    if kind == b'FM ':
        dtype = '<f4'
    elif kind == b'DM ':
        dtype = '<f8'
    else:
        raise ValueError(f'unsupported Kaldi matrix type {kind!r} (expected "FM " or "DM "; compressed and sparse '
                         f'matrices do not occur in PLDA models)')
    hdr = fd.read(10)
    s1, rows, s2, cols = struct.unpack('<bibi', hdr)
    if s1 != 4 or s2 != 4:
        raise ValueError('corrupt Kaldi matrix header')
    buf = fd.read(rows * cols * np.dtype(dtype).itemsize)
    return np.frombuffer(buf, dtype=dtype).reshape(rows, cols)


def _read_text_matrix(fd) -> np.ndarray:
    rows = []
    while True:
        line = fd.readline()
        if not line:
            raise ValueError('truncated text matrix')
        text = line.decode()
        closing = ']' in text
        vals = text.replace('[', ' ').replace(']', ' ').split()
        if vals:
            rows.append(np.array(vals, dtype=np.float32))
        if closing:
            return np.vstack(rows)


def read_plda(path_or_fd):
    """Kaldi PLDA model -> ``(mean, transform, psi)`` like kaldi_utils.py:25-54: binary ('\\0B<Plda> ' + vector +
    matrix + vector + '</Plda> ') or text."""
    fd = open(path_or_fd, 'rb') if isinstance(path_or_fd, (str, os.PathLike)) else path_or_fd
    try:
        head = fd.read(2)
        if head == b'\x00B':
            if fd.read(7) != b'<Plda> ':
                raise ValueError('not a Kaldi PLDA model')
            mean = _read_binary_vector(fd)
            trans = _read_binary_matrix(fd)
            psi = _read_binary_vector(fd)
        else:
            if head + fd.read(5) != b'<Plda> ':
                raise ValueError('not a Kaldi PLDA model')
            mean = np.array(fd.readline().decode().strip(' \n[]').split(), dtype=float)
            if fd.read(2) != b' [':
                raise ValueError('corrupt text PLDA model')
            trans = _read_text_matrix(fd)
            psi = np.array(fd.readline().decode().strip(' \n[]').split(), dtype=float)
        if fd.read(8) != b'</Plda> ':
            raise ValueError('PLDA model does not end with </Plda>')
    finally:
        if fd is not path_or_fd:
            fd.close()
    return mean, trans, psi


def write_plda(path, mean, trans, psi, dtype=np.float64):
    """Binary Kaldi PLDA model (what ``ivector-compute-plda`` writes)."""
    dt = np.dtype(dtype).newbyteorder('<')
    v, m = (b'FV ', b'FM ') if np.dtype(dtype) == np.float32 else (b'DV ', b'DM ')
    mean, trans, psi = (np.ascontiguousarray(a, dtype=dt) for a in (mean, trans, psi))
    with open(path, 'wb') as fd:
        fd.write(b'\x00B<Plda> ')
        fd.write(v + b'\x04' + struct.pack('<i', mean.shape[0]) + mean.tobytes())
        fd.write(m + struct.pack('<bibi', 4, trans.shape[0], 4, trans.shape[1]) + trans.tobytes())
        fd.write(v + b'\x04' + struct.pack('<i', psi.shape[0]) + psi.tobytes())
        fd.write(b'</Plda> ')


# ---- segments ---------------------------------------------------------------------------------------------------
def read_xvector_timing_dict(path):
    """``{recording: (array of x-vector names, array [n, 2] of start / end seconds)}`` from a Kaldi 'segments' file
    (diarization_lib.py:96-113): consecutive lines with the same recording name form one entry, a recording that
    appears again later replaces its earlier entry."""
    try:                                                          # C parser when pandas is around (100k lines: 10x)
        import pandas as pd
        tab = pd.read_csv(path, sep=r'\s+', header=None, dtype={0: str, 1: str}, engine='c', na_filter=False)
        if tab.shape[1] < 4:
            raise ValueError('segments file with fewer than four fields per line')
        names, recs = tab[0].tolist(), tab[1].tolist()
        times = tab.iloc[:, 2:].to_numpy(dtype=float)
    except ImportError:
        names, recs, times = [], [], []
        with open(path) as fd:
            for line in fd:
                f = line.split()
                if not f:
                    continue
                if len(f) < 4:
                    raise ValueError(f'segments line with fewer than four fields: {line!r}')
                names.append(f[0])
                recs.append(f[1])
                times.append([float(v) for v in f[2:]])
        times = np.array(times, dtype=float).reshape(len(names), -1)
    out = {}
    lo = 0
    for k in range(1, len(recs) + 1):
        if k == len(recs) or recs[k] != recs[lo]:
            out[recs[lo]] = (np.array(names[lo:k], dtype=object), np.array(times[lo:k], dtype=float))
            lo = k
    return out


def write_segments(path, entries):
    """entries: iterable of ``(xvector_name, recording, start, end)``."""
    with open(path, 'w') as fd:
        for name, rec, start, end in entries:
            fd.write(f'{name} {rec} {start:.3f} {end:.3f}\n')


# ---- x-vector transform -----------------------------------------------------------------------------------------
def read_xvec_transform(path):
    """``(mean1, mean2, lda)`` of vbhmm.py:125-129 from the HDF5 file of the reference's models (through h5py when
    it is installed, else through the reader of the HDF5 subset those files use) or from an ``.npz`` with the same
    three arrays."""
    if str(path).endswith('.npz'):
        with np.load(path) as z:
            return np.array(z['mean1']), np.array(z['mean2']), np.array(z['lda'])
    try:
        import h5py                                              # noqa: F401
    except ImportError:
        from .h5_minimal import read_datasets
        d = read_datasets(path, ('mean1', 'mean2', 'lda'))
        return d['mean1'], d['mean2'], d['lda']
    with h5py.File(path, 'r') as f:
        return np.array(f['mean1']), np.array(f['mean2']), np.array(f['lda'])


# ---- RTTM -------------------------------------------------------------------------------------------------------
def write_rttm(fp, file_name, labels, starts, ends):
    """One 'SPEAKER' line per segment, byte for byte what vbhmm.py:48-51 writes (``{x:03f}`` is ``%03f``: six
    decimals; labels are 1-based)."""
    for label, seg_start, seg_end in zip(labels, starts, ends):
        fp.write(f'SPEAKER {file_name} 1 {seg_start:03f} {seg_end - seg_start:03f} '
                 f'<NA> <NA> {label + 1} <NA> <NA>{os.linesep}')


def read_rttm(path):
    """``[(recording, start, duration, label)]`` of the SPEAKER lines."""
    rows = []
    with open(path) as fd:
        for line in fd:
            f = line.split()
            if f and f[0] == 'SPEAKER':
                rows.append((f[1], float(f[3]), float(f[4]), f[7]))
    return rows
