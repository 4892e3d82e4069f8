#!/usr/bin/env python
"""Register / LDS / scratch usage of every kernel of libvbx_hip.so, from hipcc's -Rpass-analysis=kernel-resource-usage
remarks (cross-compiles here, no GPU needed).  usage: tools/kernel_resources.py [substring ...] [-D...]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'vbx_amd', 'csrc')


def main():
    defs = [a for a in sys.argv[1:] if a.startswith('-D')]
    want = [a for a in sys.argv[1:] if not a.startswith('-D')]
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-pass-failed',
           '-Rpass-analysis=kernel-resource-usage', '-o', '/tmp/_kr.so', 'vbx_capi.hip'] + defs
    err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r'remark: \s*(.*?) \[-Rpass', line)
        if not m:
            continue
        txt = m.group(1).strip()
        if txt.startswith('Function Name:'):
            name = subprocess.run(['c++filt', txt.split(':', 1)[1].strip()],
                                  capture_output=True, text=True).stdout.strip()
            cur = {'name': re.sub(r'\(vbx::BatchView<\w+>\)', '', name).replace('vbx::', '').replace('void ', '')}
            rows.append(cur)
        elif cur is not None and ':' in txt:
            k, v = txt.split(':', 1)
            cur[k.strip()] = v.strip()
    print(f'{"kernel":58s} {"VGPR":>5s} {"AGPR":>5s} {"spill":>6s} {"scratch":>8s} {"occ":>4s} {"LDS":>7s}')
    for r in rows:
        if want and not any(w in r['name'] for w in want):
            continue
        print(f'{r["name"][:58]:58s} {r.get("VGPRs", "?"):>5s} {r.get("AGPRs", "?"):>5s} {r.get("VGPRs Spill", "?"):>6s} '
              f'{r.get("ScratchSize [bytes/lane]", "?"):>8s} {r.get("Occupancy [waves/SIMD]", "?"):>4s} '
              f'{r.get("LDS Size [bytes/block]", "?"):>7s}')


if __name__ == '__main__':
    main()
