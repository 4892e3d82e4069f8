"""Drop-in replacement for the reference's ``VBx/VBx.py``.

Put this directory in front of the reference's own ``VBx/`` directory on ``sys.path`` and the
unchanged driver (``from VBx import VBx``, /root/reference/VBx/vbhmm.py:45) picks up the
MI355X implementation.  See INTEGRATION.md and tools/run_vbhmm.py.
"""
import os
import sys

_repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _repo not in sys.path:
    sys.path.insert(0, _repo)

from vbx_amd.VBx import VBx, forward_backward, DER  # noqa: E402,F401
