#!/usr/bin/env python
"""Where does an iteration of a SMALL batch go: into kernels, or into the gaps between them?  From a rocprofv3 kernel trace
(rocpd database) of tools/profile_target.py on ONE stream: per kernel class the mean duration and the mean idle time on the GPU
before it starts (start - end of the previous kernel), over the steady part of the run; and the iteration as the sum of both.
A gap of a dependent launch on one queue is the command processor's (barrier bit, end-of-kernel cache work, dispatch); a gap
that grows with the host's launch cost means the host is not keeping the queue fed.
usage: launch_gaps.py <trace_results.db> [out.txt]"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r'vbx::(\w+)<([^>]*)>', name)
    return f'{m.group(1)}<{m.group(2)}>' if m else name[:50]


def main(path, out=None):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = db.execute(f'select {name_col}, start, end from kernels order by start').fetchall()
    names = [short(r[0]) for r in rows]
    loop = [k for k, n in enumerate(names) if n.startswith('chunk_loglik')]
    if len(loop) < 8:
        raise SystemExit('no iteration loop in this trace')
    first, last = loop[len(loop) // 4], loop[-2]                   # steady part: from a quarter in to the last full iteration
    dur, gap, cnt = {}, {}, {}
    for k in range(first, last):
        n = names[k]
        dur[n] = dur.get(n, 0) + rows[k][2] - rows[k][1]
        gap[n] = gap.get(n, 0) + max(0, rows[k][1] - rows[k - 1][2])
        cnt[n] = cnt.get(n, 0) + 1
    iters = sum(1 for k in loop if first <= k < last)
    wall = rows[last][1] - rows[first][1]
    lines = [f'# launch gaps from {path}: {iters} iterations, {wall / iters / 1e3:.2f} us per iteration on the GPU timeline',
             f'{"kernel":58s} {"per_iter":>8s} {"avg_us":>9s} {"gap_before_us":>14s}']
    tk = tg = 0.0
    for n in sorted(dur, key=lambda n: -dur[n]):
        lines.append(f'{n:58s} {cnt[n] / iters:8.2f} {dur[n] / cnt[n] / 1e3:9.2f} {gap[n] / cnt[n] / 1e3:14.2f}')
        tk += dur[n]
        tg += gap[n]
    lines.append(f'# per iteration: kernels {tk / iters / 1e3:.2f} us + gaps {tg / iters / 1e3:.2f} us = {(tk + tg) / iters / 1e3:.2f} us')
    text = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
