#!/usr/bin/env python
"""How well do the streams of a batch overlap?  From a rocprofv3 kernel trace (rocpd database) of tools/profile_target.py
on the library's default streams: over the steady part of the run, the share of wall time with 0 / 1 / 2 / 3+ kernels in
flight, with at least one per-chunk (streaming) kernel in flight, and the busy time per kernel class.
usage: timeline_overlap.py <trace_results.db> [skip_fraction=0.3 of the iteration loop]"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r'vbx::(\w+)<', name)
    return m.group(1) if m else name[:40]


def main(path, skip=0.3):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    qcol = next((c for c in cols if c in ('queue_id', 'queue', 'stream_id', 'stream')), None)
    rows = db.execute(f'select {name_col}, start, end' + (f', {qcol}' if qcol else '') + ' from kernels order by start').fetchall()
    chunk = [r for r in rows if short(r[0]).startswith('chunk_')]
    t0, t1 = chunk[0][1], max(r[2] for r in chunk)          # the iteration loop: first to last per-chunk kernel
    lo = t0 + skip * (t1 - t0)
    hi = t1 - 0.1 * (t1 - t0)
    ev = []
    for r in rows:
        name, s, e = short(r[0]), r[1], r[2]
        if e <= lo or s >= hi:
            continue
        s, e = max(s, lo), min(e, hi)
        stream = name.startswith('chunk_')
        ev.append((s, 1, stream))
        ev.append((e, -1, stream))
    ev.sort()
    span = hi - lo
    depth = sdepth = 0
    last = lo
    hist, shist = {}, {}
    for t, d, stream in ev:
        hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
        shist[min(sdepth, 3)] = shist.get(min(sdepth, 3), 0) + (t - last)
        last = t
        depth += d
        if stream:
            sdepth += d
    hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (hi - last)
    shist[min(sdepth, 3)] = shist.get(min(sdepth, 3), 0) + (hi - last)
    print(f'window {span / 1e6:.3f} ms of {(t1 - t0) / 1e6:.3f} ms traced; queues/streams column: {qcol}')
    print('kernels in flight      :', {k: f'{100 * v / span:.1f} %' for k, v in sorted(hist.items())})
    print('chunk kernels in flight:', {k: f'{100 * v / span:.1f} %' for k, v in sorted(shist.items())})
    busy = {}
    for r in rows:
        if r[2] > lo and r[1] < hi:
            busy[short(r[0])] = busy.get(short(r[0]), 0) + (min(r[2], hi) - max(r[1], lo))
    print('summed durations / window:', {k: f'{v / span:.2f}' for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:8]})


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.3)
