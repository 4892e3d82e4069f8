#!/usr/bin/env python
"""Where the AHC stage of a long recording goes: similarity matrix, threshold calibration, device linkage, cut.
usage: tools/ahc_stage_times.py [T ...]   (one synthetic recording per T; run on the GPU box)"""
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tools'))
import bench_driver as bd  # noqa: E402
from vbx_amd import vbhmm  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [10000, 20000]
    for T in sizes:
        with tempfile.TemporaryDirectory() as tmp:
            paths = bd.make_archive(tmp, 1, T)
            models = vbhmm.load_models(paths['transform'], paths['plda'])
            recs = vbhmm._read_recordings(paths['ark'])
            st = vbhmm.DeviceStages()
            st.project(recs, models, 128)
            capi = st._capi
            out = {'T': T}
            for rep in range(2):                      # second pass: allocator warm
                t0 = time.perf_counter()
                sc = capi.Scores.cos_similarity_resident(st._thread_ctx(), st.xv, st.row0[0], T)
                t1 = time.perf_counter()
                thr, _ = sc.two_gmm_calib(20, want_llr=False)
                t2 = time.perf_counter()
                Z = sc.linkage_average(T)
                t3 = time.perf_counter()
                lab = vbhmm.cut_linkage(Z, thr, -0.015)
                t4 = time.perf_counter()
                sc.close()
                out = {'T': T, 'cos_similarity_s': round(t1 - t0, 4), 'calibration_s': round(t2 - t1, 4),
                       'linkage_s': round(t3 - t2, 4), 'cut_s': round(t4 - t3, 4), 'clusters': int(np.max(lab)) + 1,
                       'us_per_merge': round(1e6 * (t3 - t2) / (T - 1), 2)}
            st.close()
            print(json.dumps(out))


if __name__ == '__main__':
    main()
