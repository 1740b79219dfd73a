"""Loader of the optional C++ autograd nodes (cp2_amd/csrc_torch/autograd_ext.cpp, built in-tree by
`python cp2_amd/csrc_torch/setup.py build_ext --inplace` / `__graft_entry__.build()`).

They are host-side glue only (the kernels are MIOpen / hipBLASLt / libcp2hip.so either way): without the extension the
same nodes exist as Python `torch.autograd.Function`s in cp2_amd/encoder.py, which cost more interpreter time per step.
"""
from __future__ import annotations

import ctypes
import warnings

import torch  # noqa: F401  (loads libc10_hip / libtorch_hip, which the extension links against)

_ext = None
_tried = False


def load():
    """The extension module, or None (after one warning) when it has not been built."""
    global _ext, _tried
    if _tried:
        return _ext
    _tried = True
    try:
        from . import _autograd_ext as ext
    except ImportError as err:
        warnings.warn(f"cp2_amd: C++ autograd nodes not built ({err}); using the Python nodes")
        return None
    from . import _lib
    lib = _lib.load()
    ext.set_wgrad(ctypes.cast(lib.cp2_wgrad1x1, ctypes.c_void_p).value,
                  ctypes.cast(lib.cp2_wgrad1x1_num_splits, ctypes.c_void_p).value)
    ext.set_wgrad_conv(ctypes.cast(lib.cp2_wgrad_conv, ctypes.c_void_p).value,
                       ctypes.cast(lib.cp2_wgrad_conv_num_splits, ctypes.c_void_p).value)
    ext.set_bn(ctypes.cast(lib.cp2_bn_num_partials, ctypes.c_void_p).value, ctypes.cast(lib.cp2_bn_fwd, ctypes.c_void_p).value,
               ctypes.cast(lib.cp2_bn_bwd, ctypes.c_void_p).value)
    _ext = ext
    return _ext
