"""Native-module shims with the reference's pybind11 module names.

The reference's ``models/external_function.py:7-9`` does ``import resample2d_cuda``, ``import
local_attn_reshape_cuda``, ``import block_extractor_cuda``.  ``install()`` registers three modules of
those names whose ``forward`` / ``backward`` take the same positional tensors (caller-allocated,
zero-filled outputs; return 1) and forward to the C ABI of libffwm_hip.so, so the reference's own
Python wrapper runs unmodified on MI355X.  See INTEGRATION.md.
"""
import sys

from . import block_extractor_cuda, local_attn_reshape_cuda, resample2d_cuda

_MODULES = {
    "block_extractor_cuda": block_extractor_cuda,
    "local_attn_reshape_cuda": local_attn_reshape_cuda,
    "resample2d_cuda": resample2d_cuda,
}


def install(force=False):
    """Make ``import block_extractor_cuda`` (etc.) resolve to the MI355X implementation."""
    for name, mod in _MODULES.items():
        if force or name not in sys.modules:
            sys.modules[name] = mod
    return sorted(_MODULES)
