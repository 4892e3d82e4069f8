#!/usr/bin/env python
"""Per-kernel averages of every PMC counter in a rocprofv3 rocpd database (``--pmc A B C ...`` pass).

    python tools/pmc_counters.py <results.db> [out.txt [issue.json key=value ...]]

With ``issue.json`` and the workload (the key=value words tools/profile_target.py prints) the pass is also condensed into
the file bench.py reads for ``roofline.simd_issue``: per kernel the fractions of all SIMD cycles of a launch with a matrix
instruction executing / a vector instruction issuing, stamped with the hash of the kernel sources like the PMC traffic files.

Second half of the launches of each kernel only (the first ones include first-touch effects).  Ratios between SQ
counters of one pass (e.g. SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES) are what they are good for; MI355X_MICROARCH.md
("rocprofv3 PMC slots") lists the units."""
import json
import os
import re
import sqlite3
import sys


def short(name):
    m = re.search(r'vbx::(\w+?)(?:_kernel)?<([^>]*)>', name)
    if not m:
        return name[:40]
    # the gamma write-out is an instance of chunk_post_kernel (<R, SP, REPLAY = true, SPLIT>) but not part of an iteration
    args = [a.strip() for a in m.group(2).split(',')]
    return 'chunk_post_replay' if m.group(1) == 'chunk_post' and len(args) >= 3 and args[2] == 'true' else m.group(1)


def main(path, out=None, issue_out=None, *kv):
    db = sqlite3.connect(path)
    rows = db.execute('select name, counter_name, counter_value from pmc_events').fetchall()
    agg = {}
    for name, cname, val in rows:
        agg.setdefault(short(name), {}).setdefault(cname, []).append(float(val))
    counters = sorted({c for k in agg.values() for c in k})
    lines = ['# per-launch averages (second half of the launches of each kernel) from ' + path,
             f'{"kernel":24s} {"launches":>8s} ' + ' '.join(f'{c:>26s}' for c in counters)]
    for k, d in sorted(agg.items()):
        n = max(len(v) for v in d.values())
        vals = []
        for c in counters:
            v = d.get(c, [])
            v = v[len(v) // 2:]
            vals.append(sum(v) / len(v) if v else float('nan'))
        lines.append(f'{k:24s} {n:8d} ' + ' '.join(f'{v:26.1f}' for v in vals))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in counters:
        # the counter is summed over the SIMDs of the unit a row belongs to; rows per launch x value = SIMD-cycles the
        # matrix pipes were busy; over (1024 SIMDs x launch duration x 2.4 GHz) it is the chip-wide MFMA utilisation
        cols = [r[1] for r in db.execute('pragma table_info(kernels)')]
        name_col = 'name' if 'name' in cols else [c for c in cols if 'name' in c][0]
        dur = {}
        for name, st, en in db.execute(f'select {name_col}, start, end from kernels'):
            dur.setdefault(short(name), []).append(en - st)
        lines.append('')
        lines.append('# kernel: MFMA-busy SIMD-cycles per launch, launch duration (us, under the profiler), MFMA utilisation of the')
        lines.append('# chip (busy / (1024 SIMDs x duration x 2.4 GHz)), issue activity of the resident waves (ACTIVE_INST_ANY / WAVE_CYCLES)')
        issue = {}
        for k, d in sorted(agg.items()):
            v = d.get('SQ_VALU_MFMA_BUSY_CYCLES', [])
            launches = len(dur.get(k, [])) or 1
            rows_per_launch = len(v) / launches
            v = v[len(v) // 2:]
            busy = (sum(v) / len(v) if v else 0.0) * rows_per_launch
            dd = dur.get(k, [0])
            dd = dd[len(dd) // 2:]
            us = sum(dd) / len(dd) / 1e3
            w = d.get('SQ_WAVE_CYCLES', [])
            a = d.get('SQ_ACTIVE_INST_ANY', [])
            act = (sum(a) / max(sum(w), 1.0)) if w and a else float('nan')
            util = busy / (1024 * us * 2400.0) if us > 0 else float('nan')
            line = f'{k:24s} mfma_busy_simd_cycles {busy:14.0f}   duration_us {us:9.2f}   mfma_util {util:7.4f}   active/wave_cycles {act:7.4f}'
            va = d.get('SQ_ACTIVE_INST_VALU', [])
            if va:
                # SQ_ACTIVE_INST_* count quad-cycles (4 clocks) of a SIMD issuing that class; same rows-per-launch scaling
                va = va[len(va) // 2:]
                valu = (sum(va) / len(va)) * rows_per_launch * 4.0
                vutil = valu / (1024 * us * 2400.0) if us > 0 else float('nan')
                line += f'   valu_issue_simd_cycles {valu:14.0f}   valu_util {vutil:7.4f}   valu+mfma {vutil + util:7.4f}'
                issue[k] = {'mfma_busy': util, 'valu_issue': vutil, 'duration_us_under_profiler': us, 'launches_profiled': launches}
            lines.append(line)
        if issue_out and issue:
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from vbx_amd.build import iteration_source_hash
            workload = {}
            for item in kv:
                key, val = item.split('=', 1)
                workload[key] = int(val) if val.lstrip('-').isdigit() else val
            with open(issue_out, 'w') as fh:
                json.dump({'workload': workload, 'kernels': issue, 'iteration_source_sha16': iteration_source_hash(),
                           'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU ... (one pass); mfma_busy = busy '
                                     'cycles / (1024 SIMDs x duration x 2.4 GHz), valu_issue = 4 x SQ_ACTIVE_INST_VALU / the same'},
                          fh, indent=1, sort_keys=True)
    text = '\n'.join(lines) + '\n'
    if out:
        open(out, 'w').write(text)
    print(text)


if __name__ == '__main__':
    main(*sys.argv[1:])
