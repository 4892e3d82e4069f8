#!/usr/bin/env python
"""Run the reference's UNCHANGED diarization driver ``VBx/vbhmm.py`` on top of the MI355X VBx().

    python tools/run_vbhmm.py --reference /path/to/VBx-checkout -- <vbhmm.py arguments>

``vbhmm.py`` does ``from VBx import VBx`` (vbhmm.py:45).  Run as a script it would pick up its
sibling ``VBx.py``; ``runpy.run_path`` does not prepend the script directory, so placing
``vbx_drop_in/`` in front of ``<reference>/VBx`` on ``sys.path`` makes the import resolve to this
repository's drop-in module while every other import of the driver (diarization_lib,
kaldi_utils) still comes from the reference checkout.  Nothing of the reference is copied or
modified.

``vbhmm.py`` also imports three third-party packages (``kaldi_io``, ``h5py``, ``fastcluster``;
vbhmm.py:33-35).  Where they are installed they are used as they are.  With ``--allow-shims`` any
that is missing is replaced by a minimal stand-in written to a temp directory (a Kaldi ``FV``
ark reader, a reader for the three contiguous float64 datasets of ``transform.h5``, and
``scipy.cluster.hierarchy.linkage`` for ``fastcluster.linkage``) -- enough for the ES2005a example
of ``run_example.sh:23-34``; the same stand-ins generated tests/golden/es2005a.npz.
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import runpy
import sys
import tempfile
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STAND_INS = {
    'kaldi_io': {
        'kaldi_io/__init__.py': '''
            import struct, numpy as np
            class BadSampleSize(Exception): pass
            class UnknownMatrixHeader(Exception): pass
            def open_or_fd(f, mode='rb'):
                return open(f, mode) if isinstance(f, str) else f
            def read_vec_flt_ark(path):
                """Kaldi binary vector archive: '<key> \\\\0B' + ('FV '|'DV ') + '\\\\4' + int32 n + n floats."""
                with open(path, 'rb') as fd:
                    while True:
                        key = b''
                        while True:
                            ch = fd.read(1)
                            if ch in (b'', b' '):
                                break
                            key += ch
                        if not key:
                            return
                        assert fd.read(2) == b'\\x00B', 'not a binary Kaldi archive'
                        kind = fd.read(3)
                        size = 4 if kind == b'FV ' else 8
                        assert fd.read(1) == b'\\x04'
                        n = struct.unpack('<i', fd.read(4))[0]
                        yield key.decode(), np.frombuffer(fd.read(n * size), dtype='<f4' if size == 4 else '<f8')
        ''',
        'kaldi_io/kaldi_io.py': '''
            def _read_compressed_mat(*a, **k): raise NotImplementedError('stand-in kaldi_io')
            def _read_mat_ascii(*a, **k): raise NotImplementedError('stand-in kaldi_io')
        ''',
    },
    'h5py': {
        'h5py.py': '''
            import numpy as np
            # transform.h5 of the ResNet101 models: three contiguous float64 datasets at fixed offsets
            _LAYOUT = {'mean1': (2048, (256,)), 'mean2': (4096, (128,)), 'lda': (5120, (256, 128))}
            class File:
                def __init__(self, path, mode='r'):
                    self._raw = open(path, 'rb').read()
                def __enter__(self): return self
                def __exit__(self, *a): return False
                def __getitem__(self, name):
                    off, shape = _LAYOUT[name]
                    return np.frombuffer(self._raw, dtype='<f8', count=int(np.prod(shape)), offset=off).reshape(shape)
        ''',
    },
    'fastcluster': {
        'fastcluster.py': '''
            from scipy.cluster.hierarchy import linkage as _linkage
            def linkage(y, method='single', preserve_input=True):
                return _linkage(y, method=method)
        ''',
    },
}


def write_stand_ins(root: str, names):
    for name in names:
        for rel, src in STAND_INS[name].items():
            path = os.path.join(root, rel)
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, 'w') as f:
                f.write(textwrap.dedent(src))


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--reference', default=os.environ.get('VBX_REFERENCE', '/root/reference'),
                    help='checkout of BUTSpeechFIT/VBx (the directory that contains VBx/vbhmm.py)')
    ap.add_argument('--allow-shims', action='store_true',
                    help='stand in for kaldi_io / h5py / fastcluster when they are not installed')
    ap.add_argument('rest', nargs=argparse.REMAINDER, help='-- followed by the arguments of vbhmm.py')
    args = ap.parse_args(argv)
    rest = args.rest[1:] if args.rest[:1] == ['--'] else args.rest
    script = os.path.join(args.reference, 'VBx', 'vbhmm.py')
    if not os.path.exists(script):
        raise SystemExit(f'{script} not found (pass --reference)')
    missing = [m for m in STAND_INS if importlib.util.find_spec(m) is None]
    with tempfile.TemporaryDirectory() as tmp:
        front = [os.path.join(REPO, 'vbx_drop_in')]
        if missing:
            if not args.allow_shims:
                raise SystemExit(f'vbhmm.py needs {missing}; install them or pass --allow-shims')
            write_stand_ins(tmp, missing)
            front.append(tmp)
        front.append(os.path.join(args.reference, 'VBx'))
        old_argv, old_path = sys.argv, list(sys.path)
        sys.argv = [script] + rest
        sys.path[:0] = front
        try:
            runpy.run_path(script, run_name='__main__')
        finally:
            sys.argv, sys.path[:] = old_argv, old_path


if __name__ == '__main__':
    main()
