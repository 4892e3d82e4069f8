#!/usr/bin/env python
"""profiles/<round>_c5_referee_table.md: the nine (Fa, Fb) points of BASELINE config 5 after two iterations --
|reference - truth| (tests/golden/make_golden_referee.py -> profiles/r05_c5_referee_reference.json) next to |fp64 - truth|,
|fp32 - truth|, |fp32-split - truth| of the GPU paths (tests/test_gpu_configs.py -> gpurun_out/config_parity.json, copied to
profiles/<round>_config_parity.json).  truth = oracle/vbx_oracle_x.py in numpy.longdouble.   usage: referee_table.py [round]"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = sys.argv[1] if len(sys.argv) > 1 else 'r05'
ref = json.load(open(os.path.join(REPO, 'profiles', 'r05_c5_referee_reference.json')))['points']
par = json.load(open(os.path.join(REPO, 'profiles', f'{r}_config_parity.json')))
lines = ['# BASELINE config 5, nine (Fa, Fb) points after two iterations: who is how far from the exact result',
         '',
         'truth = the reference\'s own algorithm in `numpy.longdouble` (`oracle/vbx_oracle_x.py`; log-domain and linear-domain',
         'evaluations agree to the column "referee ±").  Entries: max abs deviation of gamma (500 sampled rows) / of pi.',
         '',
         '| (Fa, Fb) | referee ± | reference (VBx.py, float64) | fp64 kernels | fp32 kernels | fp32-split kernels | reference vs fp64 kernels |',
         '|---|---|---|---|---|---|---|']
for k, v in ref.items():
    tag = f'c5/{k}/it2'
    cells = [f'{v["referee_forms_disagree"]["gamma"]:.1e}', f'**{v["gamma_rows_max_abs"]:.2e}** / {v["pi_max_abs"]:.1e}']
    for p in ('fp64', 'fp32', 'fp32-split'):
        d = par[f'{tag}/truth/{p}']
        cells.append(f'{d["gamma"]:.2e} / {d["pi"]:.1e}')
    d = par[f'{tag}/sweep9/fp64']
    cells.append(f'{d["gamma"]:.2e} / {d["pi"]:.1e}')
    fa, fb = k.split('_')
    lines.append(f'| ({fa[2:]}, {fb[2:]}) | ' + ' | '.join(cells) + ' |')
lines += ['',
          'Reading: at (.3, 64) and (.4, 64) the reference is 1.3e-4 / 1.5e-4 from the exact result of its own algorithm -- the',
          'very distance the fp64 kernels were from the reference (last column) -- while every path of this repository is within',
          'north_star\'s 1e-4 of the truth at all nine points (fp64: within 4e-8).  tests/test_gpu_configs.py holds the paths to the',
          'truth (`check_against_truth`) and reports the comparison with the reference for these two points (`REFERENCE_OFF`).']
out = os.path.join(REPO, 'profiles', f'{r}_c5_referee_table.md')
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
