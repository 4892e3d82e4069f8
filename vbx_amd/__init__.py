"""vbx_amd -- MI355X (gfx950) implementation of the VBx variational-Bayes HMM E/M loop.

Public surface (mirrors /root/reference/VBx/VBx.py):
    from vbx_amd import VBx, forward_backward, DER
plus ``vbx_amd.batch.VBx_batch`` for many independent recordings per GPU / per node.
The compute path is libvbx_hip.so (vbx_amd/csrc, C ABI in include/vbx_hip.h).
"""
from .VBx import VBx, forward_backward, DER  # noqa: F401

__all__ = ['VBx', 'forward_backward', 'DER']
__version__ = '0.1.0'
