#!/usr/bin/env python
"""One screen of a bench.py line (or several): headline, the other GEMM mode, fp64, every config.  usage: bench_summary.py file.json ..."""
import json
import sys

for f in sys.argv[1:]:
    txt = open(f).read().strip()
    d = json.loads(txt) if txt.startswith('{\n') or '\n' not in txt else json.loads(txt.splitlines()[-1])   # (the full record: bench.py --full-out)
    r = d['roofline']
    print(f'{f}: {d["gemm"]} {d["value"]:.0f} rec-it/s  {d["ms_per_step"]:.4f} ms/step | {r["kernel"]} {r["avg_launch_us"]:.1f} us frac {r["frac"]:.3f} '
          f'bound {r.get("bound_today", r["bound"])} traffic {r["traffic"]} issue {r.get("simd_issue")}')
    for key in ('f32_split', 'f32_exact', 'f64'):
        if key in d:
            o, ro = d[key], d[key]['roofline']
            print(f'   {key:9s} {o["value"]:.0f} rec-it/s  {o["ms_per_step"]:.4f} ms/step | {ro["kernel"]} {ro["avg_launch_us"]:.1f} us frac {ro["frac"]:.3f} bound {ro.get("bound_today", ro["bound"])}'
                  f' | kernels {o["kernels_avg_us"]}')
    if 'single_recording' in d:
        print('   single recording', round(d['single_recording']['ms_per_iteration'] * 1e3, 1), 'us per iteration')
    if 'cpu_baseline' in d:
        c = d['cpu_baseline']
        par = c.get('parity_over_iterations') or c.get('parity_after_2_iterations')
        print('   cpu_baseline', c['kind'], round(c['value'], 3), c['unit'], '| parity', {k: v for k, v in (par or {}).items() if k != 'note'})
    if 'call_level' in d:
        print('   call level', round(d['call_level']['ms_per_call'], 2), 'ms per VBx_batch call of', d['call_level']['iterations'], 'iterations,', round(d['call_level']['value']), 'rec-it/s')
    for k, v in d.get('configs', {}).items():
        if 'ms_per_call' in v:
            print(f'   {k:40s} {v["ms_per_call"]:.3f} ms per call, {v["iterations"]} iterations, gamma diff vs reference {v["gamma_max_abs_diff_vs_reference"]:.2e}')
        else:
            print(f'   {k:40s} {v["ms_per_iteration"]:.4f} ms/it {v["value"]:.0f} rec-it/s | {v["dominant_kernel"]} {v["avg_us"]:.1f} us frac {v["frac"]:.3f} '
                  f'bound {v.get("bound_today", v["bound"])} traffic {v["traffic"]}')
