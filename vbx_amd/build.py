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


def _find_objdump() -> str | None:
    """llvm-objdump of the ROCm in use: $ROCM_PATH, next to the hipcc that compiles, /opt/rocm, then PATH."""
    cands = []
    for root in (os.environ.get('ROCM_PATH'), os.environ.get('HIP_PATH')):
        if root:
            cands.append(os.path.join(root, 'lib', 'llvm', 'bin', 'llvm-objdump'))
    hipcc = os.environ.get('HIPCC') or shutil.which('hipcc')
    if hipcc:
        cands.append(os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), 'lib', 'llvm', 'bin', 'llvm-objdump'))
    cands += ['/opt/rocm/lib/llvm/bin/llvm-objdump', shutil.which('llvm-objdump')]
    for c in cands:
        if c and os.path.exists(c):
            return c
    return None


OBJDUMP = _find_objdump()
SOURCES = ['vbx_capi.hip']
HEADERS = ['vbx_host_state.hpp', 'vbx_host_launch.hpp', 'vbx_host_batch.hpp', 'vbx_host_group.hpp', 'vbx_host_steps.hpp', 'vbx_host_ahc.hpp', 'vbx_device.hpp', 'vbx_kernels.hpp', 'vbx_scan.hpp', 'vbx_scan_wide.hpp', 'vbx_fb_dense.hpp', 'vbx_operator.hpp', 'vbx_split.hpp', 'vbx_chunk_loglik.hpp', 'vbx_chunk_post.hpp', 'vbx_big.hpp', 'vbx_linkage.hpp', 'vbx_ahc.hpp', 'vbx_frontend.hpp', os.path.join('..', '..', 'include', 'vbx_hip.h')]
# -slp-vectorize-hor=false: the compiler's vectorised sums end in "v_pk_add_f32 d, p, p op_sel:[0,1]" (x + y of a register
# pair), one of the packed-f32 forms that misread src1 on gfx950 under matrix-instruction load (audit_isa() below,
# DESIGN section 6); with horizontal reductions left scalar none of them is generated, and audit_isa() makes sure
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-Wno-pass-failed', '-mllvm', '-slp-vectorize-hor=false']


# the sources that decide what the kernels of an EM iteration do and move: profiles/*_pmc_traffic.json carries their
# hash, and bench.py only quotes a PMC figure whose hash matches the tree it runs from
ITERATION_SOURCES = ['vbx_host_state.hpp', 'vbx_host_launch.hpp', 'vbx_host_batch.hpp', 'vbx_host_group.hpp', 'vbx_device.hpp', 'vbx_kernels.hpp', 'vbx_scan.hpp', 'vbx_operator.hpp',
                     'vbx_split.hpp', 'vbx_chunk_loglik.hpp', 'vbx_chunk_post.hpp', 'vbx_scan_wide.hpp']


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
        # a library that cannot be audited (audit_isa below) is built so that it refuses the split GEMM mode -- the one mode
        # the packed-f32 misread of DESIGN section 6 was ever seen in -- instead of accepting it unchecked
        audit = OBJDUMP is not None and not os.environ.get('VBX_AMD_SKIP_ISA_AUDIT')
        if not audit:
            import warnings
            warnings.warn('libvbx_hip.so: no ISA audit (' + ('VBX_AMD_SKIP_ISA_AUDIT is set' if OBJDUMP else 'llvm-objdump not found')
                          + '); built with -DVBX_ISA_UNAUDITED: VBX_GEMM_SPLIT / precision="fp32-split" will be refused')
        cmd = [_hipcc()] + FLAGS + ([] if audit else ['-DVBX_ISA_UNAUDITED=1']) + ['-o', tmp] + SOURCES
        if verbose:
            print(' '.join(cmd))
        res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
        if res.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError('hipcc failed:\n' + res.stdout + res.stderr)
        if audit:
            bad = audit_isa(disassemble(tmp))
            if bad:
                os.remove(tmp)
                raise RuntimeError('the compiled device code holds packed-f32 instructions that misread src1 on gfx950 '
                                   '(DESIGN section 6); restructure the source they come from:\n  ' + '\n  '.join(bad[:20]))
        os.replace(tmp, LIB)
    return LIB


def disassemble(lib: str | None = None) -> str:
    """The gfx950 code object of a built library as text (llvm-objdump; no GPU needed)."""
    import tempfile
    lib = lib or LIB
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, copy)
        subprocess.run([OBJDUMP, '--offloading', copy], cwd=tmp, check=True, capture_output=True)
        objs = [f for f in os.listdir(tmp) if 'gfx950' in f]
        if not objs:
            raise RuntimeError(f'no gfx950 code object in {lib}')
        return subprocess.run([OBJDUMP, '-d', os.path.join(tmp, objs[0])], check=True, capture_output=True, text=True).stdout


def audit_isa(asm: str) -> list[str]:
    """Instructions the device code must not contain, as 'kernel: instruction' strings.

    Packed-f32 instructions whose LOW half takes src1 from the HIGH register of its pair -- the op_sel bit of src1:
    ``v_pk_fma_f32 ... op_sel:[x,1,x]``, ``v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[x,1]``.  On gfx950 that half reads src1
    as ZERO in lanes 48-63 now and then (1e-5 ... 3e-4 of the executions) while another wavefront of the CU issues the
    K = 32 sixteen-bit matrix instructions of gfx950 (v_mfma_f32_16x16x32_f16 / _bf16: what the split GEMM mode runs); the
    selects of src0 and src2 and every op_sel_hi form are not affected, nor is anything beside the f32 / f64 / fp8 matrix
    instructions (tools/hazard/pk_opsel_probe.hip, profiles/r04_hazard/).  The compiler emits the form for the last step of a vectorised sum (x + y of a register pair:
    avoided with -slp-vectorize-hor=false) and for a multiplier it broadcasts from the odd element of a loaded vector
    (chunk_post's product at the cut in round 3).
    """
    import re
    bad, kernel = [], '?'
    sel = re.compile(r'op_sel:\[([01]),([01])')
    for line in asm.splitlines():
        if line.endswith('>:'):
            kernel = line.split('<', 1)[-1][:-2]
            continue
        code = line.split('//')[0].strip()
        if code.startswith('v_pk_') and '_f32' in code.split()[0]:
            m = sel.search(code)
            if m and m.group(2) == '1':
                bad.append(f'{kernel}: {code}')
    return bad


if __name__ == '__main__':
    print(build(force=True, verbose=True))
