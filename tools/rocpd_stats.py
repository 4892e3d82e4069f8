#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite result (``*_results.db``) into a per-kernel stats table
(calls, total / average / min / max duration) -- the same content as ``--stats`` prints, for
profiles/ when the tool was run with the default (database) output format."""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r'vbx::(\w+)<([^>]*)>', name)
    return f'{m.group(1)}<{m.group(2)}>' if m else name[:70]         # (chunk_post_kernel<.., true> = the gamma write-out)


def main(path, out=None):
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
    rows = db.execute(f'select {name_col}, start, end from kernels').fetchall()
    agg = {}
    for name, s, e in rows:
        agg.setdefault(short(name), []).append(e - s)
    total = sum(sum(v) for v in agg.values())
    lines = [f'# rocprofv3 kernel-trace summary of {path}', '# durations in microseconds',
             f'{"kernel":60s} {"calls":>7s} {"total_us":>12s} {"avg_us":>10s} {"min_us":>10s} {"max_us":>10s} {"pct":>6s}']
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        lines.append(f'{k:60s} {len(v):7d} {sum(v)/1e3:12.1f} {sum(v)/len(v)/1e3:10.2f} {min(v)/1e3:10.2f} '
                     f'{max(v)/1e3:10.2f} {100*sum(v)/total:6.2f}')
    text = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
