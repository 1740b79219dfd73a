"""CPU: the C-ABI library loads and exports every symbol include/cp2hip.h declares,
and the ops refuse to run without a GPU (no CPU fallback)."""
import pytest
import torch

from cp2_amd import _lib, ops


def test_library_loads_and_exports_header_symbols():
    lib = _lib.load()
    declared = _lib.declared_symbols()
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in cp2hip.h but not exported"
    assert set(declared) == set(_lib.SIGNATURES), "ctypes table out of sync with cp2hip.h"
    assert lib.cp2_version() >= 100
    assert lib.cp2_error_string(-1) and lib.cp2_error_string(0) == b"ok"


def test_ops_have_no_cpu_fallback():
    x = torch.zeros(2, 3, 8, 8)
    with pytest.raises(_lib.Cp2LibraryError):
        ops.compose_mask(x, x, 4)
    with pytest.raises(_lib.Cp2LibraryError):
        ops.ema_flat(torch.zeros(8), torch.zeros(8), 0.999)
    with pytest.raises(_lib.Cp2LibraryError):
        ops.enqueue(torch.zeros(4, 8), torch.zeros(2, 4), torch.zeros(1, dtype=torch.long))


def test_ema_scalars_follow_python_double_arithmetic():
    import numpy as np
    m32, om32 = ops.ema_scalars(0.999)
    assert m32 == float(np.float32(0.999))
    assert om32 == float(np.float32(1.0 - 0.999)) and om32 != float(np.float32(1.0) - np.float32(0.999))


def test_bn_partial_count_matches_library():
    lib = _lib.load()
    for M in (1, 3, 6272, 25088, 100352, 401408, 777):
        for C in (64, 128, 192, 256, 512, 1024, 2048, 4096):
            g = lib.cp2_bn_num_partials(M, C)
            assert 1 <= g <= 256 and ops._bn_partials(M, C) == g, (M, C)
    assert lib.cp2_bn_num_partials(100, 96) == -3 and lib.cp2_bn_num_partials(100, 8256) == -3
    assert lib.cp2_bn_num_partials(0, 64) == -2
