"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and exports exactly the
symbols include/pcuda_hip.h declares; argument validation fails loudly instead of launching."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "pcuda_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pcuda_[a-z0-9_]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol():
    from pointcloududa_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = _lib.lib()
    decl = _declared()
    assert len(decl) >= 45
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(handle, name), "header declares %s but the library does not export it" % name
    assert sorted(_lib.EXPORTED_SYMBOLS) == decl, "ctypes prototypes and the header disagree: %r" % (
        sorted(set(_lib.EXPORTED_SYMBOLS) ^ set(decl)),)
    hdr = int(re.search(r"#define\s+PCUDA_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "pcuda_hip.h")).read()).group(1))
    assert lib.pcuda_version() == hdr == _lib.PCUDA_ABI_VERSION


def test_binding_refuses_a_library_of_another_abi(monkeypatch):
    """pcuda_src / pcuda_dst changed layout while pcuda_version() stayed 1.  The binding now compares the
    version and every struct size when it loads the library."""
    from pointcloududa_amd import _lib
    lib = _lib.lib()
    for which, st in enumerate((_lib.ConvGeom, _lib.Src, _lib.Dst, _lib.Pooled, _lib.ReduceJob)):
        assert lib.pcuda_abi_struct_size(which) == ctypes.sizeof(st) > 0
    assert lib.pcuda_abi_struct_size(99) == 0
    assert lib.pcuda_nn_loss_workspace_floats(4, 300) == 2 * 4 * 300 + 2 * 4 * 5 and lib.pcuda_nn_loss_workspace_floats(0, 300) == 0
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "PCUDA_ABI_VERSION", _lib.PCUDA_ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="ABI version"):
        _lib.lib()
    monkeypatch.undo()
    assert _lib.lib().pcuda_version() == _lib.PCUDA_ABI_VERSION


def test_bad_arguments_are_rejected_without_a_gpu():
    from pointcloududa_amd import _lib
    from pointcloududa_amd._lib import ConvGeom, check
    lib = _lib.lib()
    g = ConvGeom(2, 8, 8, 16, 16, 15, 15, 3, 1, 1, 1, 0)            # out size inconsistent with the geometry
    assert lib.pcuda_conv2d_packed_fwd_bytes(ctypes.byref(g), 0) == 0
    rc = lib.pcuda_conv2d_pack_fwd(ctypes.byref(g), 0, None, None, None)
    assert rc == -1 and b"pack_fwd" in lib.pcuda_last_error()
    with pytest.raises(RuntimeError, match="adam_step"):
        check(lib.pcuda_adam_step(None, None, None, None, 0, 1e-3, 0.9, 0.99, 1e-8, 0.0, 1, 1.0, None), "adam_step")
    g2 = ConvGeom(2, 8, 8, 16, 16, 16, 16, 3, 1, 1, 1, 0)
    # records of the packed layout: 144 bytes (bf16x3: 32 hi | 32 lo | 8 pad bf16) against 80 (bf16: 32 values + 8 pad);
    # one 32-row tile x one 32-channel chunk x 9 taps here
    assert lib.pcuda_conv2d_packed_fwd_bytes(ctypes.byref(g2), 0) == 9 * 32 * 144
    assert lib.pcuda_conv2d_packed_fwd_bytes(ctypes.byref(g2), 1) == 9 * 32 * 80
    assert lib.pcuda_conv2d_dgrad_tiles(ctypes.byref(g2), 0) == lib.pcuda_conv2d_fwd_tiles(ctypes.byref(g2), 0)
    assert lib.pcuda_conv2d_wgrad_workspace_size(ctypes.byref(g2)) > 0
    assert lib.pcuda_conv2d_fwd_tiles(ctypes.byref(g2), 0) == 2
    assert lib.pcuda_seg_loss_workspace_size(2, 4, 256) > 0


def test_pointnet_cls_batch_of_one_raises_like_the_reference():
    """The reference's batch-1 branch cannot execute (InstanceNorm1d on the 2-D output of fc1; pinned by
    make_golden.py in param_counts.npz); the HIP module raises a RuntimeError too, before touching the device"""
    import numpy as np
    import torch
    from conftest import GOLD
    from pointcloududa_amd.networks import PointNetCls
    g = np.load(os.path.join(GOLD, "param_counts.npz"))
    assert int(g["pncls_batch1_raises"]) == 2
    with pytest.raises(RuntimeError, match="batch size 1"):
        PointNetCls()(torch.rand(1, 3, 300))
