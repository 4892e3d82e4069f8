#!/usr/bin/env python
"""HBM traffic per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected
in separate runs as MI355X_MICROARCH.md prescribes) -> profiles/<tag>_pmc_traffic.json.

    python tools/pmc_traffic.py <fetch_results.db> <write_results.db> <out.json> key=value ...

Corrections applied (MI355X_MICROARCH.md, section HBM): both counters are in KiB; on gfx950 FETCH_SIZE
reports one half of the bytes of a wide coalesced streaming read, so the read side is doubled.  WRITE_SIZE
is taken as reported (uncalibrated by the guide).  bench.py reads the JSON to fill ``roofline.traffic``
when its workload matches the one recorded here (the key=value arguments).
"""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vbx_amd.build import iteration_source_hash  # noqa: E402


def short(name):
    m = re.search(r'vbx::(\w+?)(?:_kernel)?<([^>]*)>', name)
    if not m:
        return name
    # the gamma write-out is an instance of chunk_post_kernel (<R, SP, REPLAY = true, SPLIT>) but not part of an iteration
    args = [a.strip() for a in m.group(2).split(',')]
    return 'chunk_post_replay' if m.group(1) == 'chunk_post' and len(args) >= 3 and args[2] == 'true' else m.group(1)


ITERATION_KERNELS = ('fin', 'chunk_loglik', 'scan2', 'scan_compose', 'chunk_post')     # (fin: M-step + iteration end, one launch)


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute('select name, counter_value from pmc_events where counter_name = ?', (counter,)).fetchall()
    agg = {}
    for name, val in rows:
        agg.setdefault(short(name), []).append(float(val))
    return agg


def main(fetch_db, write_db, out, *kv):
    fetch = per_kernel(fetch_db, 'FETCH_SIZE')
    write = per_kernel(write_db, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [])
        w = write.get(k, [])
        # skip the first launches (warm-up, first-touch effects): keep the second half
        f2, w2 = f[len(f) // 2:], w[len(w) // 2:]
        fk = sum(f2) / len(f2) if f2 else 0.0
        wk = sum(w2) / len(w2) if w2 else 0.0
        kernels[k] = {'launches_profiled': len(f), 'FETCH_SIZE_KiB_raw': fk, 'WRITE_SIZE_KiB_raw': wk,
                      'hbm_read_bytes': 2.0 * fk * 1024.0, 'hbm_write_bytes': wk * 1024.0,
                      'hbm_bytes_per_launch': 2.0 * fk * 1024.0 + wk * 1024.0}
    workload = {}
    for item in kv:
        key, val = item.split('=', 1)
        workload[key] = int(val) if val.lstrip('-').isdigit() else val
    iteration = sum(kernels[k]['hbm_bytes_per_launch'] for k in ITERATION_KERNELS if k in kernels)
    doc = {'iteration_hbm_bytes': iteration, 'iteration_kernels': [k for k in ITERATION_KERNELS if k in kernels],
           'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), per-launch averages',
           'corrections': 'KiB -> bytes; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported',
           'workload': workload, 'kernels': kernels,
           'iteration_source_sha16': iteration_source_hash()}     # bench.py refuses the file once the kernels have changed
    with open(out, 'w') as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    for k, v in kernels.items():
        print(f'{k:16s} read {v["hbm_read_bytes"] / 1e6:9.2f} MB  write {v["hbm_write_bytes"] / 1e6:9.2f} MB  ({v["launches_profiled"]} launches)')


if __name__ == '__main__':
    main(*sys.argv[1:])
