#!/usr/bin/env python
"""Timing of the AHC score stage (vbhmm.py:135-138) on the GPU next to the CPU oracle (NumPy restatement of
diarization_lib.cos_similarity / twoGMMcalib_lin).  One JSON line per size.  Not the headline bench."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from vbx_amd import _capi
    from vbx_amd.diarization_lib import cos_similarity, twoGMMcalib_lin
    from oracle import ahc_oracle
    ctx = _capi.default_context(0)
    rng = np.random.default_rng(0)
    for T in (1025, 10000):
        centres = rng.standard_normal((6, 128))
        x = centres[rng.integers(0, 6, T)] + 0.8 * rng.standard_normal((T, 128))
        cos_similarity(x[:64])                                        # warm-up
        t0 = time.perf_counter()
        sc = _capi.Scores.cos_similarity(ctx, x)                       # device only (H2D of x included)
        t1 = time.perf_counter()
        thr, _ = sc.two_gmm_calib(20, want_llr=False)
        t2 = time.perf_counter()
        scr = cos_similarity(x)                                        # as vbhmm.py calls it: + D2H of T*T doubles
        t3 = time.perf_counter()
        thr2, llr = twoGMMcalib_lin(scr.ravel())                       # resident matrix, + D2H of the LLRs
        t4 = time.perf_counter()
        out = {'T': T, 'D': 128, 'gpu_cos_similarity_ms': 1e3 * (t1 - t0), 'gpu_twoGMMcalib_20_ms': 1e3 * (t2 - t1),
               'api_cos_similarity_ms_with_d2h': 1e3 * (t3 - t2), 'api_twoGMMcalib_ms_with_llr_d2h': 1e3 * (t4 - t3),
               'gmm_pass_GBs': 20 * 8 * T * T / (t2 - t1) / 1e9, 'threshold': float(thr)}
        if T <= 10000:
            c0 = time.perf_counter()
            so = ahc_oracle.cos_similarity(x)
            c1 = time.perf_counter()
            to, _ = ahc_oracle.twoGMMcalib_lin(so.ravel(), niters=20 if T < 5000 else 2)
            c2 = time.perf_counter()
            out.update({'cpu_oracle_cos_similarity_ms': 1e3 * (c1 - c0),
                        'cpu_oracle_twoGMMcalib_ms_per_pass': 1e3 * (c2 - c1) / (20 if T < 5000 else 2)})
        print(json.dumps(out))
        sc.close()


if __name__ == '__main__':
    main()
