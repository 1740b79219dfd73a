"""ctypes binding of libcp2hip.so (the C ABI declared in include/cp2hip.h)."""
from __future__ import annotations

import ctypes
import os
import re
from ctypes import c_char_p, c_float, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# CP2HIP_LIB: another build of the same library (kernel experiments under tools/); it must export the same symbols
LIB_PATH = os.environ.get("CP2HIP_LIB") or os.path.join(_HERE, "lib", "libcp2hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cp2hip.h")

_P = c_void_p  # every device pointer crosses the ABI as a plain address

# name -> argument ctypes (return type is int unless listed in _RESTYPE)
SIGNATURES = {
    "cp2_version": [],
    "cp2_error_string": [c_int],
    "cp2_profile_next_launch": [_P, _P],
    "cp2_compose_mask": [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cp2_compose_pair": [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "cp2_strided_gather_f32": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "cp2_strided_gather_i64": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "cp2_gather_rows_f32": [_P, _P, _P, c_int, c_int, c_int64, _P, _P],
    "cp2_corr_iou": [_P, _P, _P, _P, _P, _P, c_int, c_int, _P],
    "cp2_corr_iou_strided": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cp2_ema_flat": [_P, _P, c_int64, c_float, c_float, _P],
    "cp2_ema_flat_timed": [_P, _P, c_int64, c_float, c_float, _P, _P, _P],
    "cp2_ema_flat_shadow": [_P, _P, _P, c_int64, c_float, c_float, _P, _P, _P],
    "cp2_ema_multi": [_P, _P, _P, _P, _P, c_int, c_float, c_float, _P],
    "cp2_enqueue": [_P, _P, _P, _P, c_int, c_int, c_int, _P],
    "cp2_feat_normalize_pool": [_P, c_int64, c_int64, c_int64, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "cp2_pool_finalize": [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "cp2_pool_bwd": [_P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, c_int, c_int, _P],
    "cp2_feat_bwd": [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int64, c_int, c_int, c_int, _P],
    "cp2_feat_normalize_pool_pair": [_P, c_int64, c_int64, c_int64, _P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P,
                                     c_int, c_int, c_int, _P],
    "cp2_feat_bwd_fused": [_P, _P, _P, _P, c_int, c_int64, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, _P, c_int64, c_int64, c_int64,
                           c_int, c_int, c_int, _P],
    "cp2_step_scalars": [_P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_float, _P, c_int, c_int, _P],
    "cp2_step_tail": [_P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_float, _P, c_int, c_int,
                      _P, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "cp2_rowkey_num_splits": [c_int, c_int],
    "cp2_rowkey_infonce_fwd": [_P, c_int, c_int64, c_int64, c_int64, c_int, _P, c_int, _P, c_int, c_float,
                               c_int, _P, _P, _P, _P, _P, c_int, c_int, _P, c_int, _P],
    "cp2_rowkey_infonce_finalize": [_P, _P, _P, _P, c_int, _P, c_int, c_float, c_float, c_int, c_int, c_int64,
                                    c_int64, c_int64, _P, _P, _P, _P, _P, _P, c_int, _P],
    "cp2_densecl_match": [_P, _P, c_int, c_int64, c_int64, c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, c_float, c_float,
                          c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cp2_dense_num_splits": [c_int, c_int],
    "cp2_dense_infonce_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, _P, _P, _P, _P,
                              _P, _P, _P, _P, _P, c_int, c_float, _P, c_int, c_int, c_int, _P],
    "cp2_loss_post": [_P, _P, _P, _P, c_int, _P, c_int, c_float, c_float, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P,
                      _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "cp2_step_post": [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P,
                      _P, _P, _P, _P, c_int, _P, c_int, c_float, c_float, c_int, c_int, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P,
                      _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P],
    "cp2_crop_resize_flip": [_P, c_int, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P],
    "cp2_pil_resize_ksize": [c_int, c_int, c_int, c_int],
    "cp2_pil_resize_workspace_bytes": [c_int, c_int, c_int, c_int, c_int],
    "cp2_pil_resize_crop": [_P, c_int, c_int, c_int, _P, _P, c_int, c_int, c_int, _P, c_int64, _P],
    "cp2_color_ops": [_P, _P, _P, c_int, c_int, c_int, _P],
    "cp2_blur_to_tensor": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cp2_erase_rect": [_P, _P, c_int, c_int, c_int, _P],
    "cp2_sgd_flat": [_P, _P, _P, _P, c_int, _P, _P, c_float, _P, c_float, c_float, _P],
    "cp2_pack_grads": [_P, _P, c_int, c_int, _P, _P, c_float, _P],
    "cp2_bf16_image": [_P, _P, c_int64, _P],
    "cp2_wgrad1x1_num_splits": [c_int, c_int, c_int],
    "cp2_wgrad1x1": [_P, _P, _P, _P, c_int, c_int, c_int, _P],
    "cp2_wgrad_conv_num_splits": [c_int] * 7,
    "cp2_wgrad_conv": [_P, _P, _P, _P] + [c_int] * 12 + [_P],
    "cp2_bn_num_partials": [c_int, c_int],
    "cp2_bn_fwd": [_P, _P, _P, _P, _P, _P, c_float, c_float, c_int, _P, _P, _P, _P, _P, c_int, c_int, _P],
    "cp2_bn_bwd": [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, _P],
    "cp2_masked_quantiles": [_P, c_int64, c_int64, c_int, c_int, _P, _P, c_int, c_int, _P, c_int, _P, _P, c_int64, _P],
    "cp2_quantiles_workspace_bytes": [c_int, _P, _P, c_int],
    "cp2_masked_quantiles_multi": [c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int64, _P],
    "cp2_maxpool3s2_fwd": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cp2_maxpool3s2_bwd": [_P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cp2_cutpaste": [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "cp2_mirror_loss_num_partials": [c_int, c_int64],
    "cp2_mirror_loss": [_P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int64, _P],
    "cp2_dense_infonce_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, c_float, c_float, c_float, c_float, _P, _P, c_float,
                              _P, _P, c_int, c_float, _P, c_int, c_int, c_int, _P],
}
_RESTYPE = {"cp2_error_string": c_char_p, "cp2_quantiles_workspace_bytes": c_int64, "cp2_pil_resize_workspace_bytes": c_int64}

_lib = None


class Cp2LibraryError(RuntimeError):
    pass


def declared_symbols(header: str = HEADER_PATH):
    """Names of every function declared in include/cp2hip.h."""
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cp2_[a-z0-9_]+)\s*\(", text)))


def load() -> ctypes.CDLL:
    """Load libcp2hip.so or fail loudly -- there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Cp2LibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C cp2_amd/csrc`).  cp2_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise Cp2LibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().cp2_error_string(rc)
        raise Cp2LibraryError(f"{what} failed with code {rc}: {msg.decode() if msg else '?'}")
