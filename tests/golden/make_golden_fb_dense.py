#!/usr/bin/env python
"""forward_backward() known answers for ARBITRARY transition matrices, from the reference itself
(/root/reference/VBx/VBx.py:146-175); authoring container only.  -> tests/golden/fb_dense_cases.npz"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden.make_golden import ref_module  # noqa: E402
from vbx_amd.synth import make_lls  # noqa: E402


def main():
    ref = ref_module()
    out = {}
    for name, (T, S, seed, kind) in {'dense_T64_S5': (64, 5, 0, 'dirichlet'), 'dense_T300_S31': (300, 31, 1, 'dirichlet'),
                                     'sparse_T200_S12': (200, 12, 2, 'sparse'), 'left_right_T150_S8': (150, 8, 3, 'left_right'),
                                     'dense_T120_S70': (120, 70, 4, 'dirichlet'), 'dense_T40_S130': (40, 130, 5, 'dirichlet'),
                                     'one_frame_S6': (1, 6, 6, 'dirichlet')}.items():
        rng = np.random.default_rng(100 + seed)
        lls, _ = make_lls(T, S, seed=seed)
        if kind == 'dirichlet':
            tr = rng.dirichlet(np.full(S, 0.3), size=S)
        elif kind == 'sparse':                       # most transitions impossible (only the eps of VBx.py:158 lets them through)
            tr = rng.dirichlet(np.full(S, 0.3), size=S) * (rng.random((S, S)) < 0.3)
            tr += np.eye(S) * 0.2
            tr /= tr.sum(1, keepdims=True)
        else:                                        # left-to-right chain
            tr = np.zeros((S, S))
            for i in range(S):
                tr[i, i] = 0.9
                tr[i, min(i + 1, S - 1)] += 0.1
        ip = rng.dirichlet(np.ones(S))
        post, tll, lfw, lbw = ref.forward_backward(lls, tr, ip)
        for k, v in dict(lls=lls, tr=tr, ip=ip, post=post, tll=np.asarray(tll), lfw=lfw, lbw=lbw).items():
            out[f'{name}/{k}'] = v
        print(name, T, S, float(tll))
    np.savez_compressed(os.path.join(HERE, 'fb_dense_cases.npz'), **out)


if __name__ == '__main__':
    main()
