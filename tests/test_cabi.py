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


def test_split_geometry_functions_are_pure_host_code():
    """Split counts the callers size their workspaces with (no GPU needed: plain host arithmetic in the library)."""
    lib = _lib.load()
    assert lib.cp2_dense_num_splits(8, 4096) == 2          # BASELINE config 4: 256 (sample, tile) items -> 512 workgroups
    assert lib.cp2_dense_num_splits(32, 196) == 4          # bench shape: at most one 64-pixel tile per split
    assert lib.cp2_dense_num_splits(8, 1024) == 8
    assert lib.cp2_dense_num_splits(2, 16) == 1 and lib.cp2_dense_num_splits(64, 4096) == 1
    assert lib.cp2_dense_num_splits(0, 16) == -2
    assert lib.cp2_wgrad1x1_num_splits(6272, 2048, 512) == 8 and lib.cp2_wgrad1x1_num_splits(100352, 64, 64) == 448
    assert lib.cp2_wgrad1x1_num_splits(32, 64, 64) == 1
    assert lib.cp2_wgrad1x1_num_splits(6272, 96, 64) == -3 and lib.cp2_wgrad1x1_num_splits(0, 64, 64) == -2


def test_sgd_flat_plan_covers_every_slot_once():
    """Block table of cp2_sgd_flat: blocks never straddle tensors, cover each tensor's elements exactly once, and
    address the flat buffer at the slot offsets MODEL.flatten_parameters() hands out (256-byte aligned slots)."""
    import numpy as np
    import torch
    numels = [64, 9408, 1, 513, 36864, 7]
    offs, total = [], 0
    for n in numels:
        offs.append(total)
        total += (n + 63) // 64 * 64
    plan = ops.SgdFlatPlan(offs, numels, torch.device("cpu"))
    tab = plan.blk_tab.numpy()
    first = np.ctypeslib.as_array(plan.first)
    assert plan.ntensors == len(numels) and first[0] == 0 and first[-1] == len(tab)
    for t, (off, n) in enumerate(zip(offs, numels)):
        rows = tab[first[t]:first[t + 1]]
        assert (rows[:, 0] == t).all()
        assert (rows[:, 3] > 0).all() and (rows[:, 3] <= ops.SGD_BLOCK_FLOATS).all() and rows[:, 3].sum() == n
        assert (rows[:, 1] - off == rows[:, 2]).all() and (rows[:, 2] == np.arange(len(rows)) * ops.SGD_BLOCK_FLOATS).all()


def test_cpp_autograd_nodes_are_built_and_importable():
    """cp2_amd/_autograd_ext (host-side C++ autograd nodes) is part of the build: it must import and be wired to the
    library's weight-gradient entry points."""
    from cp2_amd import _cext
    ext = _cext.load()
    assert ext is not None, "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    assert all(hasattr(ext, n) for n in ("conv1x1", "shadow_weight", "set_wgrad"))


def test_pack_grads_and_fused_bn_node_entry_points():
    """cp2_pack_grads rejects bad arguments before anything is launched (host-side checks only: safe without a GPU), and the
    C++ extension exports the fused-BN node next to the convolution nodes."""
    import ctypes
    lib = _lib.load()
    ptrs = (ctypes.c_void_p * 2)()
    first = (ctypes.c_int32 * 3)(0, 1, 2)
    assert lib.cp2_pack_grads(None, ptrs, 0, 2, 64, first, 1.0, None) == -1                # null flat buffer
    assert lib.cp2_pack_grads(64, None, 0, 2, 64, first, 1.0, None) == -1
    assert lib.cp2_pack_grads(64, ptrs, 2, 2, 64, first, 1.0, None) == -2                  # empty tensor range
    assert lib.cp2_pack_grads(64, ptrs, -1, 2, 64, first, 1.0, None) == -2
    assert lib.cp2_pack_grads(68, ptrs, 0, 2, 64, first, 1.0, None) == -4                  # flat buffer not 16-byte aligned
    from cp2_amd import _cext
    ext = _cext.load()
    assert ext is not None and hasattr(ext, "fused_bn") and hasattr(ext, "set_bn")
