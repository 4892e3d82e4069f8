"""Drop-in for the reference's ``VBx/diarization_lib.py``: every name of the reference module, with
``cos_similarity`` and ``twoGMMcalib_lin`` (the T*T score stage of the AHC initialisation,
vbhmm.py:135-138) replaced by the MI355X implementations.

Put this directory in front of the reference's ``VBx/`` directory on ``sys.path`` (tools/run_vbhmm.py
does); the reference module is then loaded from the next ``diarization_lib.py`` found on the path.
"""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_repo = os.path.dirname(_here)
if _repo not in sys.path:
    sys.path.insert(0, _repo)

_ref_path = None
for _d in sys.path:
    _cand = os.path.join(_d or '.', 'diarization_lib.py')
    if os.path.isfile(_cand) and os.path.abspath(os.path.dirname(_cand)) != _here:
        _ref_path = _cand
        break
if _ref_path is None:
    raise ImportError('vbx_drop_in/diarization_lib.py: the reference diarization_lib.py is not on sys.path')
_spec = importlib.util.spec_from_file_location('_reference_diarization_lib', _ref_path)
_ref = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_ref)
globals().update({k: v for k, v in vars(_ref).items() if not k.startswith('__')})

from vbx_amd.diarization_lib import cos_similarity, twoGMMcalib_lin  # noqa: E402,F401
