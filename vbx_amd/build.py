"""Build libvbx_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

The shared object lands next to its sources (vbx_amd/csrc/libvbx_hip.so) so that it travels
with a snapshot of the repository; it is git-ignored.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.environ.get('VBX_AMD_LIB') or os.path.join(CSRC, 'libvbx_hip.so')   # (VBX_AMD_LIB: an experiment build, tools/build_variants.sh)
SOURCES = ['vbx_capi.hip']
HEADERS = ['vbx_device.hpp', 'vbx_kernels.hpp', 'vbx_scan.hpp', 'vbx_scan_wide.hpp', 'vbx_fb_dense.hpp', 'vbx_operator.hpp', 'vbx_split.hpp', 'vbx_chunk_loglik.hpp', 'vbx_chunk_post.hpp', 'vbx_linkage.hpp', 'vbx_ahc.hpp', 'vbx_frontend.hpp', os.path.join('..', '..', 'include', 'vbx_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-pass-failed']


# the sources that decide what the kernels of an EM iteration do and move: profiles/*_pmc_traffic.json carries their
# hash, and bench.py only quotes a PMC figure whose hash matches the tree it runs from
ITERATION_SOURCES = ['vbx_capi.hip', 'vbx_device.hpp', 'vbx_kernels.hpp', 'vbx_scan.hpp', 'vbx_operator.hpp',
                     'vbx_split.hpp', 'vbx_chunk_loglik.hpp', 'vbx_chunk_post.hpp']


def iteration_source_hash() -> str:
    import hashlib
    h = hashlib.sha256()
    for f in ITERATION_SOURCES:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(f.encode() + b'\0' + fh.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found (need ROCm >= 7.0 to build libvbx_hip.so for gfx950)')


def is_stale() -> bool:
    if os.environ.get('VBX_AMD_LIB'):
        return False
    if not os.path.exists(LIB):
        return True
    built = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > built for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the library if it is missing or older than its sources; return its path."""
    if not force and not is_stale():
        return LIB
    # every rank of a torchrun job may find the library stale at once: one builds (file lock), the others wait and
    # then find it fresh; the output goes to a name of its own and is moved into place atomically
    import fcntl
    with open(os.path.join(CSRC, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not is_stale():
            return LIB
        tmp = f'{LIB}.{os.getpid()}.tmp'
        cmd = [_hipcc()] + FLAGS + ['-o', tmp] + SOURCES
        if verbose:
            print(' '.join(cmd))
        res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
        if res.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)
        os.replace(tmp, LIB)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
