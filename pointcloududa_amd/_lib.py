"""ctypes binding of libpcuda_hip.so (the C ABI declared in include/pcuda_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call
returns an error code, a RuntimeError is raised with the library's own message.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCUDA_LIB") or os.path.join(_HERE, "lib", "libpcuda_hip.so")   # (PCUDA_LIB: A/B builds)

PREC_BF16X3, PREC_BF16 = 0, 1
ACT_SIGMOID, ACT_SOFTMAX = 0, 1
FAM_CONV_FWD, FAM_CONV_WGRAD, FAM_POINTWISE, FAM_DENSE_F32 = 0, 1, 2, 3

c_f32p = C.c_void_p      # all device pointers travel as integers
i32, i64, f32, vp, sz = C.c_int, C.c_longlong, C.c_float, C.c_void_p, C.c_size_t


class ConvGeom(C.Structure):
    _fields_ = [("n", i32), ("cin", i32), ("cout", i32), ("in_h", i32), ("in_w", i32), ("out_h", i32),
                ("out_w", i32), ("k", i32), ("stride", i32), ("pad", i32), ("dil", i32), ("in_up", i32)]


class ReduceJob(C.Structure):      # pcuda_reduce_job (include/pcuda_hip.h)
    _fields_ = [("partial", C.c_void_p), ("numel", C.c_longlong), ("dw", C.c_void_p), ("db_partial", C.c_void_p),
                ("nb", C.c_longlong), ("db", C.c_void_p), ("ksplit", C.c_int), ("nkg", C.c_int), ("accumulate", C.c_int),
                ("ntaps", C.c_int)]


class Src(C.Structure):
    _fields_ = [("p1", vp), ("sn1", i64), ("sc1", i64), ("scale1", vp), ("shift1", vp),
                ("p2", vp), ("sn2", i64), ("sc2", i64), ("scale2", vp), ("shift2", vp), ("c1", i32)]


class Dst(C.Structure):
    _fields_ = [("p1", vp), ("sn1", i64), ("sc1", i64), ("p2", vp), ("sn2", i64), ("sc2", i64), ("c1", i32)]


class Pooled(C.Structure):
    _fields_ = [("g", vp), ("g_sn", i64), ("g_sc", i64), ("g2", vp), ("g2_sn", i64), ("g2_sc", i64), ("idx", vp),
                ("h", i32), ("w", i32)]


_PROTOS = {
    "pcuda_version": (i32, []),
    "pcuda_device_count": (i32, []),
    "pcuda_last_error": (C.c_char_p, []),
    "pcuda_build_hash": (C.c_char_p, []),
    "pcuda_launch_count": (i64, [i32]),
    "pcuda_fallback_count": (i64, []),
    "pcuda_abi_struct_size": (sz, [i32]),
    "pcuda_last_kernel": (C.c_char_p, []),
    "pcuda_clock_probe": (i32, [vp, vp, i32, vp]),
    "pcuda_prof_enable": (i32, [i32]),
    "pcuda_prof_reset": (i32, []),
    "pcuda_prof_read": (i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64)]),
    "pcuda_prof_dump": (i32, [C.c_char_p]),
    "pcuda_debug_read_clocks": (i32, [vp]),
    "pcuda_conv2d_packed_fwd_bytes": (sz, [C.POINTER(ConvGeom), i32]),
    "pcuda_conv2d_packed_dgrad_bytes": (sz, [C.POINTER(ConvGeom), i32]),
    "pcuda_conv2d_pack_fwd": (i32, [C.POINTER(ConvGeom), i32, vp, vp, vp]),
    "pcuda_conv2d_pack_all": (i32, [C.POINTER(ConvGeom), i32, vp, vp, vp, vp]),
    "pcuda_conv2d_pack_dgrad": (i32, [C.POINTER(ConvGeom), i32, vp, vp, vp]),
    "pcuda_conv2d_pack_job_bytes": (sz, []),
    "pcuda_conv2d_pack_jobs_fill": (i32, [C.POINTER(ConvGeom), i32, vp, vp, vp, vp, i32, C.POINTER(i32)]),
    "pcuda_conv2d_pack_table": (i32, [vp, vp, i32, i32, vp]),
    "pcuda_conv2d_fwd_tiles": (i32, [C.POINTER(ConvGeom), i32]),
    "pcuda_conv2d_forward": (i32, [C.POINTER(ConvGeom), i32, C.POINTER(Src), vp, vp, f32, C.POINTER(Dst), vp, vp]),
    "pcuda_conv2d_d1_forward_ok": (i32, [C.POINTER(ConvGeom)]),
    "pcuda_conv2d_dgrad": (i32, [C.POINTER(ConvGeom), i32, C.POINTER(Src), vp, C.POINTER(Dst), i32, vp]),
    "pcuda_conv2d_dgrad_tiles": (i32, [C.POINTER(ConvGeom), i32]),
    "pcuda_conv2d_dgrad_bnred": (i32, [C.POINTER(ConvGeom), i32, C.POINTER(Src), vp, C.POINTER(Dst), i32, vp, i64, i64, vp,
                                       vp, vp, vp]),
    "pcuda_conv2d_dgrad_fold": (i32, [C.POINTER(ConvGeom), i32, C.POINTER(Src), vp, C.POINTER(Dst), vp, i64, i64, vp, vp, vp, vp]),
    "pcuda_conv2d_dgrad_lrelu": (i32, [C.POINTER(ConvGeom), i32, C.POINTER(Src), vp, C.POINTER(Dst), vp, i64, i64, f32, vp]),
    "pcuda_conv2d_wgrad_workspace_size": (sz, [C.POINTER(ConvGeom)]),
    "pcuda_conv2d_wgrad": (i32, [C.POINTER(ConvGeom), i32, C.POINTER(Src), vp, i64, i64, vp, vp, i32, vp, sz, vp]),
    "pcuda_conv2d_wgrad_partial": (i32, [C.POINTER(ConvGeom), i32, C.POINTER(Src), vp, i64, i64, vp, vp, i32, vp, sz,
                                         C.POINTER(ReduceJob), vp]),
    "pcuda_wgrad_reduce_batch": (i32, [C.POINTER(ReduceJob), i32, vp]),
    "pcuda_bn_finalize": (i32, [vp, i32, i32, i64, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp]),
    "pcuda_bn_stats": (i32, [vp, i64, i64, i32, i32, i64, vp, C.POINTER(i32), vp]),
    "pcuda_bn_apply": (i32, [vp, i64, i64, vp, vp, i32, vp, i64, i64, i32, i32, i64, vp]),
    "pcuda_bn_bwd_reduce": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, vp, i32, i32, i32, i64, vp,
                                  C.POINTER(i32), vp]),
    "pcuda_bn_bwd_finalize": (i32, [vp, i32, i32, i64, vp, vp, vp, vp, vp, i32, vp, vp]),
    "pcuda_bn_bwd_apply": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, i32, f32, vp, i64, i64, i32,
                                 i32, i64, vp]),
    "pcuda_lrelu_bwd": (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, f32, vp, i64, i64, i32, i32, i64, vp]),
    "pcuda_channel_sum": (i32, [vp, i64, i64, i32, i32, i64, vp, i32, vp, sz, vp]),
    "pcuda_maxpool2_fwd": (i32, [vp, i64, i64, vp, vp, vp, i64, i64, vp, i32, i32, i32, i32, vp]),
    "pcuda_maxpool2_bwd": (i32, [vp, i64, i64, vp, i64, i64, vp, vp, i64, i64, i32, i32, i32, i32, i32, vp]),
    "pcuda_bn_bwd_reduce_pooled": (i32, [C.POINTER(Pooled), vp, i64, i64, vp, i64, i64, vp, vp, i32, i32, vp, C.POINTER(i32), vp]),
    "pcuda_bn_bwd_apply_pooled": (i32, [C.POINTER(Pooled), vp, i64, i64, vp, i64, i64, vp, f32, vp, i64, i64, i32, i32, vp]),
    "pcuda_lrelu_bwd_pooled": (i32, [C.POINTER(Pooled), vp, i64, i64, vp, i64, i64, f32, vp, i64, i64, i32, i32, vp]),
    "pcuda_upsample2_bwd": (i32, [vp, i64, i64, vp, i64, i64, i32, i32, i32, i32, i32, vp]),
    "pcuda_upsample2_bwd_bnred": (i32, [vp, i64, i64, vp, i64, i64, i32, vp, i64, i64, vp, vp, vp, C.POINTER(i32), i32, i32,
                                        i32, i32, vp]),
    "pcuda_bilinear_fwd": (i32, [vp, i64, i64, i32, i32, i32, i32, vp, i32, i32, vp]),
    "pcuda_bilinear_bwd": (i32, [vp, i32, i32, i32, i32, vp, i64, i64, i32, i32, vp]),
    "pcuda_unfold_taps": (i32, [vp, i64, i64, i32, i32, i32, i32, i32, i32, i32, i32, vp, i32, i32, vp]),
    "pcuda_add4": (i32, [vp, vp, vp, vp, vp, i64, vp]),
    "pcuda_mul": (i32, [vp, vp, vp, i64, vp]),
    "pcuda_entropy_fwd": (i32, [vp, i32, f32, vp, vp, i32, i32, i64, vp]),
    "pcuda_entropy_bwd": (i32, [vp, i32, f32, vp, vp, vp, i32, i32, i32, i64, vp]),
    "pcuda_entropy_bwd2": (i32, [vp, i32, f32, vp, vp, vp, vp, i32, i32, i32, i64, vp]),
    "pcuda_sum_all_workspace_size": (sz, []),
    "pcuda_sum_all": (i32, [vp, i64, C.c_double, vp, vp, sz, vp]),
    "pcuda_seg_loss_workspace_size": (sz, [i32, i32, i64]),
    "pcuda_seg_loss_fwd": (i32, [vp, vp, i32, i32, i32, i64, vp, vp, sz, vp]),
    "pcuda_seg_loss_bwd": (i32, [vp, vp, i32, i32, i32, i64, vp, vp, vp, vp, vp]),
    "pcuda_jaccard_workspace_size": (sz, [i32]),
    "pcuda_jaccard_fwd": (i32, [vp, vp, i32, i32, i32, i64, f32, vp, vp, sz, vp]),
    "pcuda_jaccard_bwd": (i32, [vp, i32, i32, i32, i64, f32, vp, vp, vp, vp]),
    "pcuda_bce_const_fwd": (i32, [vp, i64, f32, vp, vp, vp]),
    "pcuda_bce_const_bwd": (i32, [vp, i64, f32, vp, f32, vp, vp]),
    "pcuda_nn_loss_workspace_floats": (sz, [i32, i32]),
    "pcuda_nn_loss_fwd": (i32, [vp, vp, i32, i32, vp, vp, vp, vp]),
    "pcuda_nn_loss_bwd": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp]),
    "pcuda_dice_metric": (i32, [vp, vp, i32, i32, i64, vp, vp, sz, vp]),
    "pcuda_assemble_batch": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "pcuda_argmax_labels": (i32, [vp, i32, i64, i64, i32, i32, i64, vp, vp]),
    "pcuda_label_dice": (i32, [vp, vp, i64, i32, vp, vp, sz, vp]),
    "pcuda_linear_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, vp]),
    "pcuda_linear_bwd_x": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "pcuda_linear_bwd_w_workspace_size": (sz, [i32, i32, i32]),
    "pcuda_linear_bwd_w": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
    "pcuda_conv1d_k1_fwd_tiles": (i32, [i32, i32]),
    "pcuda_conv1d_k1_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "pcuda_conv1d_k1_dgrad": (i32, [vp, vp, vp, i32, i32, i32, i32, vp]),
    "pcuda_conv1d_k1_wgrad_workspace_size": (sz, [i32, i32, i32, i32]),
    "pcuda_conv1d_k1_wgrad": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
    "pcuda_max_points_fwd": (i32, [vp, i32, i32, i32, vp, vp, vp]),
    "pcuda_max_points_bwd": (i32, [vp, vp, i32, i32, i32, vp, vp]),
    "pcuda_bmm": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "pcuda_surface_vertices": (i32, [vp, i32, i32, i32, vp, i32, vp, vp, sz, vp]),
    "pcuda_surface_vertices_mc": (i32, [vp, i32, i32, i32, vp, i32, vp, vp]),
    "pcuda_fps": (i32, [vp, vp, vp, i32, i32, i32, vp, vp]),
    "pcuda_adam_step": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, f32, vp]),
    "pcuda_adam_step_dev": (i32, [vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, vp, f32, vp]),
    "pcuda_sgd_step": (i32, [vp, vp, vp, i64, f32, f32, f32, i32, f32, vp]),
}

EXPORTED_SYMBOLS = tuple(_PROTOS.keys())
_lib = None


PCUDA_E_UNSUPPORTED = -2      # include/pcuda_hip.h
PCUDA_ABI_VERSION = 5         # include/pcuda_hip.h: the header this binding was written against


def lib():
    """The loaded library; raises if it has not been built (run `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libpcuda_hip.so is missing at %s: build it with pointcloududa_amd/csrc/Makefile "
                               "(there is no CPU fallback)" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(handle, name)      # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        # a stale library (or header) would read these structs from our memory with another layout: refuse it
        ver = handle.pcuda_version()
        if ver != PCUDA_ABI_VERSION:
            raise RuntimeError("libpcuda_hip.so at %s has ABI version %d, this binding is written against %d: rebuild it "
                               "(python -c 'import __graft_entry__ as g; g.build()')" % (LIB_PATH, ver, PCUDA_ABI_VERSION))
        for which, st in enumerate((ConvGeom, Src, Dst, Pooled, ReduceJob)):
            if handle.pcuda_abi_struct_size(which) != C.sizeof(st):
                raise RuntimeError("libpcuda_hip.so: sizeof(%s) is %d in the library, %d in the binding: header / library mismatch"
                                   % (st.__name__, handle.pcuda_abi_struct_size(which), C.sizeof(st)))
        _lib = handle
    return _lib


def source_hash(csrc: str) -> str:
    """sha256[:16] over the kernel sources (csrc/*.hip, *.h, Makefile and include/pcuda_hip.h; names and bytes, sorted).
    The Makefile runs this same function to stamp the library (build/srchash.h -> pcuda_build_hash())."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) +
                   [os.path.join(csrc, "Makefile"), os.path.join(os.path.dirname(os.path.dirname(csrc)), "include", "pcuda_hip.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def csrc_hash() -> str:
    """The hash the LOADED library was compiled from (``pcuda_build_hash``): what a committed profile records next to its
    numbers, so that a figure measured on another build is never quoted as this build's (bench.py refuses a traffic
    summary whose hash differs).  ``tree_hash()`` is the same recipe over the sources on disk, when they are there."""
    v = lib().pcuda_build_hash()
    return v.decode() if v else "unknown"


def tree_hash():
    """hash of the sources on disk, or None when the package is installed without them"""
    try:
        return source_hash(os.path.join(_HERE, "csrc"))
    except OSError:
        return None


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().pcuda_last_error()
        raise RuntimeError("libpcuda_hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
