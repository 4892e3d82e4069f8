#!/usr/bin/env python
"""oracle/_ref/: the reference's own VBx.py, for the `cpu_baseline` leg of bench.py (kind "reference") -- TEST INFRASTRUCTURE.

    python oracle/build_ref.py        (also run by __graft_entry__.build())

Where /root/reference exists (the authoring container) the unmodified /root/reference/VBx/VBx.py is copied to
oracle/_ref/VBx_reference.py.  oracle/_ref/ is git-ignored -- reference sources never enter this repository's history --
but not gpurun-ignored, so the copy travels to the GPU box with the snapshot, like the built libvbx_hip.so.  The file
needs NumPy and scipy.special.logsumexp only (requirements.txt:1-2 of the reference).  Nothing under vbx_amd/ imports it:
bench.py's cpu_baseline times it, nothing else.
"""
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = '/root/reference/VBx/VBx.py'
DST = os.path.join(HERE, '_ref', 'VBx_reference.py')


def build(verbose=True):
    if not os.path.exists(SRC):
        if verbose:
            print('oracle/_ref: /root/reference is not here -- keeping', DST if os.path.exists(DST) else 'nothing (bench.py will time the port)')
        return os.path.exists(DST)
    os.makedirs(os.path.dirname(DST), exist_ok=True)
    shutil.copyfile(SRC, DST)
    if verbose:
        print('oracle/_ref:', SRC, '->', DST)
    return True


if __name__ == '__main__':
    sys.exit(0 if build() else 1)
