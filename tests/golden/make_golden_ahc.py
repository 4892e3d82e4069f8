#!/usr/bin/env python
"""Golden fixtures for the AHC score stage, generated FROM THE REFERENCE (authoring container only).

Imports /root/reference/VBx/diarization_lib.py (numpy + scipy only) and records, for the ES2005a
x-vectors after the transform of vbhmm.py:125-129 and for two synthetic sets, the inputs and the
reference outputs of cos_similarity() and twoGMMcalib_lin().  Large outputs are stored as checksums +
a sample of entries; the inputs are small and stored whole.
    tests/golden/ahc_cases.npz
"""
import importlib.util
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'


def ref_lib():
    spec = importlib.util.spec_from_file_location('_ref_diarization_lib', f'{REF}/VBx/diarization_lib.py')
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def read_ark(path):
    out = []
    with open(path, 'rb') as fd:
        while True:
            key = b''
            while True:
                ch = fd.read(1)
                if ch in (b'', b' '):
                    break
                key += ch
            if not key:
                break
            assert fd.read(2) == b'\x00B' and fd.read(3) == b'FV ' and fd.read(1) == b'\x04'
            n = struct.unpack('<i', fd.read(4))[0]
            out.append(np.frombuffer(fd.read(4 * n), dtype='<f4'))
    return np.array(out)


def es2005a_x(lib):
    raw = open(f'{REF}/VBx/models/ResNet101_16kHz/transform.h5', 'rb').read()
    mean1 = np.frombuffer(raw, '<f8', 256, 2048)
    mean2 = np.frombuffer(raw, '<f8', 128, 4096)
    lda = np.frombuffer(raw, '<f8', 256 * 128, 5120).reshape(256, 128)
    x = read_ark(f'{REF}/exp/ES2005a.ark')
    return lib.l2_norm(lda.T.dot((lib.l2_norm(x - mean1)).transpose()).transpose() - mean2)     # vbhmm.py:129


def main():
    lib = ref_lib()
    rng = np.random.default_rng(0)
    cases = {'es2005a': es2005a_x(lib)}
    centres = rng.standard_normal((4, 64))
    cases['synth_T400_D64'] = centres[rng.integers(0, 4, 400)] + 0.7 * rng.standard_normal((400, 64))
    cases['synth_T130_D20'] = rng.standard_normal((130, 20)) * rng.random((130, 1)) * 5
    out = {}
    for name, x in cases.items():
        scr = lib.cos_similarity(x)
        thr, llr = lib.twoGMMcalib_lin(scr.ravel())
        thr5, llr5 = lib.twoGMMcalib_lin(scr.ravel(), niters=5)
        idx = rng.integers(0, scr.size, 2000)
        out[name + '/x'] = x
        out[name + '/scr_sample_idx'] = idx
        out[name + '/scr_sample'] = scr.ravel()[idx]
        out[name + '/scr_stats'] = np.array([scr.sum(), (scr ** 2).sum(), scr.min(), scr.max(), np.trace(scr)])
        out[name + '/thr'] = np.array(thr)
        out[name + '/thr5'] = np.array(thr5)
        out[name + '/llr_sample'] = llr[idx]
        out[name + '/llr_stats'] = np.array([llr.sum(), (llr ** 2).sum(), llr.min(), llr.max()])
        print(name, x.shape, 'thr', thr, 'thr5', thr5)
    np.savez_compressed(os.path.join(HERE, 'ahc_cases.npz'), **out)


if __name__ == '__main__':
    main()
