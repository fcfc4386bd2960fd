"""torch-tensor front end of the C ABI (ctypes calls on torch's current HIP stream).

torch is used here for device memory (``torch.empty``) and the stream handle only; every
arithmetic step is a kernel of libpcuda_hip.so.  There is no fallback: a tensor that is not
on a HIP device, or a missing library, raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref
from typing import Optional, Tuple

import torch

from . import _lib as L
from ._lib import ACT_SIGMOID, ACT_SOFTMAX, PREC_BF16, PREC_BF16X3, ConvGeom, Dst, Src, check

_PREC = {"bf16x3": PREC_BF16X3, "bf16": PREC_BF16}
_precision = _PREC.get(os.environ.get("PCUDA_PRECISION", "bf16x3").lower(), PREC_BF16X3)


def set_precision(name: str) -> None:
    """'bf16x3' (parity mode: split-bf16 MFMA, ~fp32 accuracy) or 'bf16' (throughput mode)."""
    global _precision
    _precision = _PREC[name.lower()]


def get_precision() -> str:
    return "bf16x3" if _precision == PREC_BF16X3 else "bf16"


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, dtype=torch.float32):
    if not t.is_cuda:
        raise RuntimeError("pointcloududa_amd kernels need HIP device tensors (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError("expected %s, got %s" % (dtype, t.dtype))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _planes(t: torch.Tensor) -> Tuple[int, int, int, int, int]:
    """(n, c, hw, sn, sc) of a [N,C,...] tensor whose trailing dims form a dense plane."""
    _req(t)
    n, c = t.shape[0], t.shape[1]
    hw = 1
    for d in t.shape[2:]:
        hw *= d
    exp = 1
    for d in range(t.dim() - 1, 1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            raise ValueError("tensor planes must be dense (shape %s strides %s)" % (tuple(t.shape), t.stride()))
        exp *= t.shape[d]
    return n, c, hw, t.stride(0), t.stride(1)


class TA:
    """A tensor with an optional per-channel affine still to be applied on load
    (= a BatchNorm output that was never materialised)."""
    __slots__ = ("t", "scale", "shift")

    def __init__(self, t, scale=None, shift=None):
        self.t, self.scale, self.shift = t, scale, shift


def _as_ta(x):
    return x if isinstance(x, TA) else TA(x)


def make_src(a, b=None) -> Src:
    a = _as_ta(a)
    _, c1, _, sn1, sc1 = _planes(a.t)
    s = Src()
    s.p1, s.sn1, s.sc1, s.scale1, s.shift1, s.c1 = a.t.data_ptr(), sn1, sc1, _ptr(a.scale), _ptr(a.shift), c1
    if b is not None:
        b = _as_ta(b)
        _, _, _, sn2, sc2 = _planes(b.t)
        s.p2, s.sn2, s.sc2, s.scale2, s.shift2 = b.t.data_ptr(), sn2, sc2, _ptr(b.scale), _ptr(b.shift)
    return s


def make_dst(a: torch.Tensor, b: Optional[torch.Tensor] = None) -> Dst:
    _, c1, _, sn1, sc1 = _planes(a)
    d = Dst()
    d.p1, d.sn1, d.sc1, d.c1 = a.data_ptr(), sn1, sc1, c1
    if b is not None:
        _, _, _, sn2, sc2 = _planes(b)
        d.p2, d.sn2, d.sc2 = b.data_ptr(), sn2, sc2
    return d


# ------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------
# PCUDA_BNRED=0: the BatchNorm-backward reduce of a block's first BatchNorm as its own kernel instead of riding in the
# epilogue of the second convolution's dgrad (A/B switch)
_fuse_bnred = os.environ.get("PCUDA_BNRED", "1") != "0"
# PCUDA_FOLD_DGRAD=0: the 2x2 fold behind an up-convolution's data gradient as its own kernel (upsample2_bwd) again (A/B switch)
_fold_dgrad = os.environ.get("PCUDA_FOLD_DGRAD", "1") != "0"


def upsample_red_only(dx, bnred):
    """(dx, None): the caller computes the BatchNorm-backward reduce of ``dx`` itself (bn_backward without ``red``)"""
    return dx, None


_conv_ops = weakref.WeakSet()      # every ConvOp alive (repack_owner finds a network's layers through their ``owner``)


class ConvOp:
    """Geometry + packed-weight cache of one nn.Conv2d (square kernel, groups=1)."""

    def __init__(self, cin, cout, k, stride=1, pad=0, dil=1, in_up=False):
        self.cin, self.cout, self.k, self.stride, self.pad, self.dil, self.in_up = cin, cout, k, stride, pad, dil, in_up
        self._pk = {}     # (kind, prec) -> (weight ptr, version, generation, packed tensor)
        self._last_fwd = None   # (weight tensor, geometry) of the latest forward-layout pack: what repack_owner replays
        _conv_ops.add(self)
        # training: a forward-layout repack is followed by a dgrad in the same step, so both go out in one launch;
        # set False for layers whose input never needs a gradient / for inference-only use
        self.pack_dgrad_with_fwd = os.environ.get("PCUDA_PACK_ALL", "1") != "0"
        self.owner = None  # object whose ``_wgen`` counter is bumped when weights change behind torch's back

    def out_hw(self, in_h, in_w):
        f = lambda v: (v + 2 * self.pad - self.dil * (self.k - 1) - 1) // self.stride + 1
        return f(in_h), f(in_w)

    def geom(self, n, in_h, in_w) -> ConvGeom:
        """in_h/in_w: LOGICAL input size (already doubled when in_up)."""
        oh, ow = self.out_hw(in_h, in_w)
        return ConvGeom(n, self.cin, self.cout, in_h, in_w, oh, ow, self.k, self.stride, self.pad, self.dil,
                        1 if self.in_up else 0)

    def _packed(self, kind: str, w: torch.Tensor, g: ConvGeom):
        key = (kind, _precision)
        hit = self._pk.get(key)
        gen = getattr(self.owner, "_wgen", 0)
        if hit is not None and hit[0] == w.data_ptr() and hit[1] == w._version and hit[2] == gen:
            return hit[3]
        _req(w)
        if not w.is_contiguous():
            raise ValueError("conv weight must be contiguous OIHW")
        lib = L.lib()
        if kind == "fwd":
            nbytes = lib.pcuda_conv2d_packed_fwd_bytes(C.byref(g), _precision)
        else:
            nbytes = lib.pcuda_conv2d_packed_dgrad_bytes(C.byref(g), _precision)
        if nbytes == 0:
            raise RuntimeError("conv geometry rejected by libpcuda_hip")
        buf = hit[3] if (hit is not None and hit[3].numel() == nbytes) else torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        if kind == "fwd" and self.pack_dgrad_with_fwd:
            # the weights changed (optimiser step): repack the forward layout AND the dgrad layouts in one launch
            dkey = ("dgrad", _precision)
            dhit = self._pk.get(dkey)
            dbytes = lib.pcuda_conv2d_packed_dgrad_bytes(C.byref(g), _precision)
            dbuf = dhit[3] if (dhit is not None and dhit[3].numel() == dbytes) else torch.empty(
                dbytes, dtype=torch.uint8, device=w.device)
            check(lib.pcuda_conv2d_pack_all(C.byref(g), _precision, w.data_ptr(), buf.data_ptr(), dbuf.data_ptr(),
                                            _stream()), "pack_all")
            self._pk[dkey] = (w.data_ptr(), w._version, gen, dbuf)
        else:
            fn = lib.pcuda_conv2d_pack_fwd if kind == "fwd" else lib.pcuda_conv2d_pack_dgrad
            check(fn(C.byref(g), _precision, w.data_ptr(), buf.data_ptr(), _stream()), "pack_" + kind)
        self._pk[key] = (w.data_ptr(), w._version, gen, buf)
        if kind == "fwd":
            self._last_fwd = (w, g)
        return buf

    def forward(self, x, w, b, slope, in_h, in_w, x2=None, out=None, want_stats=False):
        """x (and x2 for a channel concat): tensors or TA.  Returns (y, partials|None, ntiles)."""
        xa = _as_ta(x)
        n = xa.t.shape[0]
        g = self.geom(n, in_h, in_w)
        if out is None:
            out = torch.empty((n, self.cout, g.out_h, g.out_w), dtype=torch.float32, device=w.device)
        pk = self._packed("fwd", w, g)
        lib = L.lib()
        partials, nt = None, 0
        if want_stats:
            nt = lib.pcuda_conv2d_fwd_tiles(C.byref(g), _precision)
            partials = torch.empty((nt, self.cout, 2), dtype=torch.float32, device=w.device)
        src, dst = make_src(xa, x2), make_dst(out)
        check(lib.pcuda_conv2d_forward(C.byref(g), _precision, C.byref(src), pk.data_ptr(), _ptr(b), float(slope),
                                       C.byref(dst), _ptr(partials), _stream()), "conv2d_forward")
        return out, partials, nt

    def dgrad(self, dy, w, in_h, in_w, dx=None, dx2=None, accumulate=False, bnred=None):
        """dy: [n,cout,oh,ow].  dx (+dx2 split along channels) = gradient of the logical input.
        ``bnred=(a, BNState)``: the BatchNorm-backward reduce of the layer in front of this convolution rides in the
        epilogue; returns (dx, (partials, ntiles)) -- or (dx, None) where the geometry does not allow it."""
        n = dy.shape[0]
        g = self.geom(n, in_h, in_w)
        if dx is None:
            dx = torch.empty((n, self.cin, in_h, in_w), dtype=torch.float32, device=dy.device)
        pk = self._packed("dgrad", w, g)
        src, dst = make_src(dy), make_dst(dx, dx2)
        if bnred is not None:
            a, st = bnred
            lib = L.lib()
            nt = lib.pcuda_conv2d_dgrad_tiles(C.byref(g), _precision) if (_fuse_bnred and dx2 is None) else 0
            if nt > 0:
                _, _, _, asn, asc = _planes(a)
                red = torch.empty((nt, self.cin, 2), dtype=torch.float32, device=dy.device)
                rc = lib.pcuda_conv2d_dgrad_bnred(C.byref(g), _precision, C.byref(src), pk.data_ptr(), C.byref(dst),
                                                  1 if accumulate else 0, a.data_ptr(), asn, asc, st.mean.data_ptr(),
                                                  st.invstd.data_ptr(), red.data_ptr(), _stream())
                if rc == 0:
                    return dx, (red, nt)
                if rc != L.PCUDA_E_UNSUPPORTED:
                    check(rc, "conv2d_dgrad_bnred")
            check(lib.pcuda_conv2d_dgrad(C.byref(g), _precision, C.byref(src), pk.data_ptr(), C.byref(dst),
                                         1 if accumulate else 0, _stream()), "conv2d_dgrad")
            return dx, None
        check(L.lib().pcuda_conv2d_dgrad(C.byref(g), _precision, C.byref(src), pk.data_ptr(), C.byref(dst),
                                         1 if accumulate else 0, _stream()), "conv2d_dgrad")
        return dx

    def dgrad_lrelu(self, dy, w, in_h, in_w, a, slope):
        """lrelu_bwd(dgrad(dy), a, slope) in one kernel where the plan allows it (pcuda_conv2d_dgrad_lrelu): the LeakyReLU
        backward of the layer in front rides in the data-gradient epilogue (GAN.py:97-108 going back)"""
        if _fuse_lrelu_dgrad and self.k > 1 and a.is_contiguous():
            n = dy.shape[0]
            g = self.geom(n, in_h, in_w)
            dz = torch.empty((n, self.cin, in_h, in_w), dtype=torch.float32, device=dy.device)
            pk = self._packed("dgrad", w, g)
            src, dst = make_src(dy), make_dst(dz)
            _, _, _, asn, asc = _planes(a)
            rc = L.lib().pcuda_conv2d_dgrad_lrelu(C.byref(g), _precision, C.byref(src), pk.data_ptr(), C.byref(dst),
                                                  a.data_ptr(), asn, asc, float(slope), _stream())
            if rc == 0:
                return dz
            if rc != L.PCUDA_E_UNSUPPORTED:
                check(rc, "conv2d_dgrad_lrelu")
        return lrelu_bwd(self.dgrad(dy, w, in_h, in_w), a, slope)

    def dgrad_fold(self, dy, w, in_h, in_w, bnred=None):
        """The data gradient of an ``in_up`` layer at the STORED (half) resolution: dgrad + the 2x2 fold of the nearest-x2
        backward in one kernel (pcuda_conv2d_dgrad_fold); ``bnred=(a, BNState)`` as in ``dgrad``.  Returns (dx_half, (partials,
        ntiles) | None) like ``upsample2_bwd(dgrad(...), bnred)``, which it falls back to where the plan does not fold."""
        assert self.in_up
        n = dy.shape[0]
        g = self.geom(n, in_h, in_w)
        lib = L.lib()
        if _fold_dgrad and in_h % 2 == 0 and in_w % 4 == 0:
            pk = self._packed("dgrad", w, g)
            dx = torch.empty((n, self.cin, in_h // 2, in_w // 2), dtype=torch.float32, device=dy.device)
            src, dst = make_src(dy), make_dst(dx)
            a_p = m_p = i_p = r_p = None
            asn = asc = 0
            red, nt = None, 0
            if bnred is not None and _fuse_bnred:
                a, st = bnred
                nt = lib.pcuda_conv2d_dgrad_tiles(C.byref(g), _precision)
                if nt > 0:
                    _, _, _, asn, asc = _planes(a)
                    red = torch.empty((nt, self.cin, 2), dtype=torch.float32, device=dy.device)
                    a_p, m_p, i_p, r_p = a.data_ptr(), st.mean.data_ptr(), st.invstd.data_ptr(), red.data_ptr()
            rc = lib.pcuda_conv2d_dgrad_fold(C.byref(g), _precision, C.byref(src), pk.data_ptr(), C.byref(dst), a_p, asn, asc,
                                             m_p, i_p, r_p, _stream())
            if rc == 0:
                if bnred is None:
                    return dx
                return (dx, (red, nt)) if red is not None else upsample_red_only(dx, bnred)
            if rc != L.PCUDA_E_UNSUPPORTED:
                check(rc, "conv2d_dgrad_fold")
        d_up = self.dgrad(dy, w, in_h, in_w)
        return upsample2_bwd(d_up, bnred=bnred)

    def wgrad(self, x, dy, dw, db, in_h, in_w, x2=None, accumulate=True):
        xa = _as_ta(x)
        n = xa.t.shape[0]
        g = self.geom(n, in_h, in_w)
        lib = L.lib()
        ws_bytes = lib.pcuda_conv2d_wgrad_workspace_size(C.byref(g))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device)
        _, _, _, sn, sc = _planes(dy)
        src = make_src(xa, x2)
        pend = getattr(_deferred, "jobs", None)
        if pend is not None:
            # inside ``deferred_wgrad_reduces()``: only the partial sums now; the split-K reduce of every layer of the
            # pass goes out in one launch when the block ends (the workspace lives in the pending list until then)
            job = L.ReduceJob()
            check(lib.pcuda_conv2d_wgrad_partial(C.byref(g), _precision, C.byref(src), dy.data_ptr(), sn, sc, dw.data_ptr(),
                                                 _ptr(db), 1 if accumulate else 0, ws.data_ptr(), ws_bytes, C.byref(job),
                                                 _stream()), "conv2d_wgrad_partial")
            if job.ksplit > 0:
                if any(j.dw == job.dw for j, _ in pend):
                    flush_wgrad_reduces()        # two gradients into one buffer: keep their order
                pend.append((job, ws))
                if _batch_reduce_n >= 2 and len(pend) >= _batch_reduce_n:
                    flush_wgrad_reduces()
            return
        check(lib.pcuda_conv2d_wgrad(C.byref(g), _precision, C.byref(src), dy.data_ptr(), sn, sc, dw.data_ptr(),
                                     _ptr(db), 1 if accumulate else 0, ws.data_ptr(), ws_bytes, _stream()),
              "conv2d_wgrad")


# PCUDA_BATCH_REDUCE=1: the split-K reduces of a backward pass in one launch per 56 layers instead of one per layer.
# OFF by default -- measured SLOWER (51.7 vs 49.5 ms per step, same box): a layer's partial sums (~35 MB) are still in
# the 256 MB infinity cache when its own reduce follows at once, and the workspace block is reused by the next layer; all
# of a pass's partials together (~1.4 GB) go out to HBM and come back.  The launches were not the cost.
# PCUDA_BATCH_REDUCE=k (k >= 2): flush every k layers -- the pending slabs (k x ~35 MB) stay inside the infinity cache.
_batch_reduce_n = int(os.environ.get("PCUDA_BATCH_REDUCE", "0") or 0)
_batch_reduce = _batch_reduce_n >= 1
_deferred = threading.local()


def flush_wgrad_reduces():
    """Run the pending split-K reduces of this thread's ``deferred_wgrad_reduces`` block (one launch per 56 layers).
    Call before anything reads the weight gradients inside the block (e.g. the all-reduce hook of the backward pass)."""
    pend = getattr(_deferred, "jobs", None)
    if not pend:
        return
    arr = (L.ReduceJob * len(pend))(*[j for j, _ in pend])
    check(L.lib().pcuda_wgrad_reduce_batch(arr, len(pend), _stream()), "wgrad_reduce_batch")
    del pend[:]          # (stream order keeps the workspaces valid: the allocator reuses them behind the reduce)


class deferred_wgrad_reduces:
    """``with deferred_wgrad_reduces():`` around one backward pass of a network (one stream, each weight once)."""

    def __enter__(self):
        self.outer = getattr(_deferred, "jobs", None)
        if _batch_reduce and self.outer is None:
            _deferred.jobs = []
        return self

    def __exit__(self, *exc):
        if _batch_reduce and self.outer is None:
            try:
                if exc[0] is None:
                    flush_wgrad_reduces()
            finally:
                _deferred.jobs = None
        return False


_batch_repack = os.environ.get("PCUDA_PACK_TABLE", "1") != "0"


def repack_owner(owner):
    """All packed-weight layouts of ``owner``'s convolutions in ONE launch (called behind the optimiser kernel that
    changed the weights).  The jobs -- raw pointers of the flat master weights and of the packed buffers, which both
    stay put -- are collected once into a device table and replayed every step; layers that have not run yet (no
    packed buffers) keep the lazy per-layer path.  Lazily, the segmenter's 43 repack launches sat in the dependent
    chain of the next forward pass."""
    if not _batch_repack:
        return False
    lib = L.lib()
    ops = [op for op in _conv_ops if op.owner is owner and op._last_fwd is not None]
    if not ops:
        return False
    gen = getattr(owner, "_wgen", 0)
    key, entries = [], []
    for op in ops:
        w, g = op._last_fwd
        fhit = op._pk.get(("fwd", _precision))
        dhit = op._pk.get(("dgrad", _precision)) if op.pack_dgrad_with_fwd else None
        if fhit is None or fhit[0] != w.data_ptr() or (op.pack_dgrad_with_fwd and dhit is None):
            continue
        entries.append((op, w, g, fhit[3], None if dhit is None else dhit[3]))
        key.append((id(op), w.data_ptr(), fhit[3].data_ptr(), 0 if dhit is None else dhit[3].data_ptr()))
    if not entries:
        return False
    key = (tuple(sorted(key)), _precision)
    tab = getattr(owner, "_pack_table", None)
    if tab is None or tab[0] != key:
        jb = lib.pcuda_conv2d_pack_job_bytes()
        chunks, blocks = [], []
        for op, w, g, fb, db in entries:
            host = C.create_string_buffer(jb * 8)
            jblk = (C.c_int * 8)()
            nj = lib.pcuda_conv2d_pack_jobs_fill(C.byref(g), _precision, w.data_ptr(), fb.data_ptr(),
                                                 None if db is None else db.data_ptr(), host, 8, jblk)
            if nj < 0:
                check(nj, "pack_jobs_fill")
            chunks.append(host.raw[:jb * nj]); blocks += list(jblk[:nj])
        dev = entries[0][1].device
        blob = torch.frombuffer(bytearray(b"".join(chunks)), dtype=torch.uint8).to(dev)
        first = torch.tensor([sum(blocks[:j]) for j in range(len(blocks))], dtype=torch.int32).to(dev)
        tab = (key, blob, first, len(blocks), sum(blocks))
        owner._pack_table = tab
    check(lib.pcuda_conv2d_pack_table(tab[1].data_ptr(), tab[2].data_ptr(), tab[3], tab[4], _stream()), "pack_table")
    for op, w, g, fb, db in entries:          # the caches are current for this weight generation
        op._pk[("fwd", _precision)] = (w.data_ptr(), w._version, gen, fb)
        if db is not None:
            op._pk[("dgrad", _precision)] = (w.data_ptr(), w._version, gen, db)
    return True


# ------------------------------------------------------------------------------------------
# BatchNorm pieces
# ------------------------------------------------------------------------------------------
class BNState:
    """per-forward statistics of one BatchNorm layer"""
    __slots__ = ("mean", "invstd", "scale", "shift", "count")


def bn_finalize(partials, ntiles, count, gamma, beta, running_mean, running_var, eps=1e-5, momentum=0.1) -> BNState:
    c = partials.shape[1]
    st = BNState()
    buf = torch.empty((4, c), dtype=torch.float32, device=partials.device)
    st.mean, st.invstd, st.scale, st.shift, st.count = buf[0], buf[1], buf[2], buf[3], count
    check(L.lib().pcuda_bn_finalize(partials.data_ptr(), ntiles, c, count, _ptr(gamma), _ptr(beta), eps, momentum,
                                    _ptr(running_mean), _ptr(running_var), st.mean.data_ptr(), st.invstd.data_ptr(),
                                    st.scale.data_ptr(), st.shift.data_ptr(), _stream()), "bn_finalize")
    return st


def bn_stats(a: torch.Tensor):
    n, c, hw, sn, sc = _planes(a)
    nt = C.c_int(0)
    lib = L.lib()
    check(lib.pcuda_bn_stats(None, sn, sc, n, c, hw, None, C.byref(nt), _stream()), "bn_stats(query)")
    partials = torch.empty((nt.value, c, 2), dtype=torch.float32, device=a.device)
    check(lib.pcuda_bn_stats(a.data_ptr(), sn, sc, n, c, hw, partials.data_ptr(), C.byref(nt), _stream()), "bn_stats")
    return partials, nt.value, n * hw


def bn_apply(a: torch.Tensor, st: BNState, relu=False, out=None):
    n, c, hw, sn, sc = _planes(a)
    if out is None:
        out = torch.empty_like(a, memory_format=torch.contiguous_format)
    _, _, _, osn, osc = _planes(out)
    check(L.lib().pcuda_bn_apply(a.data_ptr(), sn, sc, st.scale.data_ptr(), st.shift.data_ptr(), 1 if relu else 0,
                                 out.data_ptr(), osn, osc, n, c, hw, _stream()), "bn_apply")
    return out


def bn_backward(dy, a, st: BNState, gamma, dgamma, dbeta, dy2=None, post_relu=False, act_slope=1.0,
                accumulate=True, red=None, frozen=False):
    """Backward of [a = lrelu(z, act_slope)] -> BN (post_relu=False) or a -> BN -> ReLU (post_relu=True).
    Returns dz (gradient w.r.t. the pre-activation conv output, or w.r.t. a when post_relu).
    ``frozen``: ``st`` holds the running statistics (eval-mode BatchNorm): the layer is a fixed affine."""
    n, c, hw, asn, asc = _planes(a)
    _, _, _, dsn, dsc = _planes(dy)
    d2p, d2sn, d2sc = None, 0, 0
    if dy2 is not None:
        _, _, _, d2sn, d2sc = _planes(dy2)
        d2p = dy2.data_ptr()
    lib = L.lib()
    pr = 1 if post_relu else 0
    if red is not None:          # (partials, ntiles) from the producing dgrad's epilogue (ConvOp.dgrad(bnred=...))
        assert dy2 is None and not post_relu
        red, ntv = red
    else:
        nt = C.c_int(0)
        check(lib.pcuda_bn_bwd_reduce(None, 0, 0, None, 0, 0, None, 0, 0, None, None, None, None, 0, n, c, hw, None,
                                      C.byref(nt), _stream()), "bn_bwd_reduce(query)")
        red = torch.empty((nt.value, c, 2), dtype=torch.float32, device=a.device)
        check(lib.pcuda_bn_bwd_reduce(dy.data_ptr(), dsn, dsc, d2p, d2sn, d2sc, a.data_ptr(), asn, asc,
                                      st.mean.data_ptr(), st.invstd.data_ptr(), st.scale.data_ptr(), st.shift.data_ptr(),
                                      pr, n, c, hw, red.data_ptr(), C.byref(nt), _stream()), "bn_bwd_reduce")
        ntv = nt.value
    coef = torch.empty((c, 3), dtype=torch.float32, device=a.device)
    check(lib.pcuda_bn_bwd_finalize(red.data_ptr(), ntv, c, -(n * hw) if frozen else n * hw, _ptr(gamma), st.invstd.data_ptr(),
                                    st.mean.data_ptr(), _ptr(dgamma), _ptr(dbeta), 1 if accumulate else 0,
                                    coef.data_ptr(), _stream()), "bn_bwd_finalize")
    dz = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    _, _, _, zsn, zsc = _planes(dz)
    check(lib.pcuda_bn_bwd_apply(dy.data_ptr(), dsn, dsc, d2p, d2sn, d2sc, a.data_ptr(), asn, asc, coef.data_ptr(),
                                 st.scale.data_ptr(), st.shift.data_ptr(), pr, float(act_slope), dz.data_ptr(), zsn,
                                 zsc, n, c, hw, _stream()), "bn_bwd_apply")
    return dz


def make_pooled(g, idx, h, w, g2=None):
    """gradient arriving through a 2x2 max-pool, read in place by the kernel that consumes it (pcuda_pooled)"""
    _, _, _, gsn, gsc = _planes(g)
    p = L.Pooled()
    p.g, p.g_sn, p.g_sc = g.data_ptr(), gsn, gsc
    if g2 is not None:
        _, _, _, g2sn, g2sc = _planes(g2)
        p.g2, p.g2_sn, p.g2_sc = g2.data_ptr(), g2sn, g2sc
    assert idx.is_contiguous() and idx.dtype == torch.uint8
    p.idx, p.h, p.w = idx.data_ptr(), h, w
    return p


_fuse_pool = os.environ.get("PCUDA_FUSE_POOL_BWD", "1") != "0"
_fuse_lrelu_dgrad = os.environ.get("PCUDA_FUSE_LRELU_DGRAD", "1") != "0"


def lrelu_bwd_pooled(g, idx, a, slope, g2=None, dy=None):
    """lrelu_bwd(maxpool2_bwd(g (+ g2), idx), a, dy2=dy) without the full-resolution intermediate"""
    n, c, hw, asn, asc = _planes(a)
    h, w = a.shape[2], a.shape[3]
    if not _fuse_pool:
        return lrelu_bwd(maxpool2_bwd(g, idx, h, w, dy2=g2), a, slope, dy2=dy)
    dp, dsn, dsc = None, 0, 0
    if dy is not None:
        _, _, _, dsn, dsc = _planes(dy)
        dp = dy.data_ptr()
    dz = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    _, _, _, zsn, zsc = _planes(dz)
    pool = make_pooled(g, idx, h, w, g2)
    check(L.lib().pcuda_lrelu_bwd_pooled(C.byref(pool), dp, dsn, dsc, a.data_ptr(), asn, asc, float(slope), dz.data_ptr(),
                                         zsn, zsc, n, c, _stream()), "lrelu_bwd_pooled")
    return dz


def bn_backward_pooled(g, idx, a, st: BNState, gamma, dgamma, dbeta, g2=None, dy=None, act_slope=1.0, accumulate=True,
                       frozen=False):
    """bn_backward(maxpool2_bwd(g (+ g2), idx), a, ..., dy2=dy) without the full-resolution intermediate"""
    n, c, hw, asn, asc = _planes(a)
    h, w = a.shape[2], a.shape[3]
    if not _fuse_pool:
        return bn_backward(maxpool2_bwd(g, idx, h, w, dy2=g2), a, st, gamma, dgamma, dbeta, dy2=dy, act_slope=act_slope,
                           accumulate=accumulate, frozen=frozen)
    dp, dsn, dsc = None, 0, 0
    if dy is not None:
        _, _, _, dsn, dsc = _planes(dy)
        dp = dy.data_ptr()
    lib = L.lib()
    pool = make_pooled(g, idx, h, w, g2)
    nt = C.c_int(0)
    check(lib.pcuda_bn_bwd_reduce_pooled(C.byref(pool), None, 0, 0, None, 0, 0, None, None, n, c, None, C.byref(nt),
                                         _stream()), "bn_bwd_reduce_pooled(query)")
    red = torch.empty((nt.value, c, 2), dtype=torch.float32, device=a.device)
    check(lib.pcuda_bn_bwd_reduce_pooled(C.byref(pool), dp, dsn, dsc, a.data_ptr(), asn, asc, st.mean.data_ptr(),
                                         st.invstd.data_ptr(), n, c, red.data_ptr(), C.byref(nt), _stream()),
          "bn_bwd_reduce_pooled")
    coef = torch.empty((c, 3), dtype=torch.float32, device=a.device)
    check(lib.pcuda_bn_bwd_finalize(red.data_ptr(), nt.value, c, -(n * hw) if frozen else n * hw, _ptr(gamma),
                                    st.invstd.data_ptr(), st.mean.data_ptr(), _ptr(dgamma), _ptr(dbeta),
                                    1 if accumulate else 0, coef.data_ptr(), _stream()), "bn_bwd_finalize")
    dz = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    _, _, _, zsn, zsc = _planes(dz)
    check(lib.pcuda_bn_bwd_apply_pooled(C.byref(pool), dp, dsn, dsc, a.data_ptr(), asn, asc, coef.data_ptr(),
                                        float(act_slope), dz.data_ptr(), zsn, zsc, n, c, _stream()), "bn_bwd_apply_pooled")
    return dz


def lrelu_bwd(dy, a, slope, dy2=None):
    n, c, hw, asn, asc = _planes(a)
    _, _, _, dsn, dsc = _planes(dy)
    d2p, d2sn, d2sc = None, 0, 0
    if dy2 is not None:
        _, _, _, d2sn, d2sc = _planes(dy2)
        d2p = dy2.data_ptr()
    dz = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    _, _, _, zsn, zsc = _planes(dz)
    check(L.lib().pcuda_lrelu_bwd(dy.data_ptr(), dsn, dsc, d2p, d2sn, d2sc, a.data_ptr(), asn, asc, float(slope),
                                  dz.data_ptr(), zsn, zsc, n, c, hw, _stream()), "lrelu_bwd")
    return dz


def channel_sum(dz, db, accumulate=True):
    n, c, hw, sn, sc = _planes(dz)
    ws = torch.empty((n * ((hw + 2047) // 2048), c), dtype=torch.float32, device=dz.device)
    check(L.lib().pcuda_channel_sum(dz.data_ptr(), sn, sc, n, c, hw, db.data_ptr(), 1 if accumulate else 0,
                                    ws.data_ptr(), ws.numel() * 4, _stream()), "channel_sum")


# ------------------------------------------------------------------------------------------
# pooling / resampling / adds
# ------------------------------------------------------------------------------------------
def maxpool2_fwd(x):
    xa = _as_ta(x)
    n, c, _, sn, sc = _planes(xa.t)
    h, w = xa.t.shape[2], xa.t.shape[3]
    y = torch.empty((n, c, h // 2, w // 2), dtype=torch.float32, device=xa.t.device)
    idx = torch.empty((n, c, h // 2, w // 2), dtype=torch.uint8, device=xa.t.device)
    check(L.lib().pcuda_maxpool2_fwd(xa.t.data_ptr(), sn, sc, _ptr(xa.scale), _ptr(xa.shift), y.data_ptr(),
                                     y.stride(0), y.stride(1), idx.data_ptr(), n, c, h, w, _stream()), "maxpool2_fwd")
    return y, idx


def maxpool2_bwd(dy, idx, h, w, dy2=None):
    n, c, _, dsn, dsc = _planes(dy)
    d2p, d2sn, d2sc = None, 0, 0
    if dy2 is not None:
        _, _, _, d2sn, d2sc = _planes(dy2)
        d2p = dy2.data_ptr()
    dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
    check(L.lib().pcuda_maxpool2_bwd(dy.data_ptr(), dsn, dsc, d2p, d2sn, d2sc, idx.data_ptr(), dx.data_ptr(),
                                     dx.stride(0), dx.stride(1), 0, n, c, h, w, _stream()), "maxpool2_bwd")
    return dx


def upsample2_bwd(dy, bnred=None):
    """2x2 fold (backward of nearest x2).  ``bnred=(a, BNState)``: the BatchNorm-backward reduce of the layer that
    consumes the result rides along; returns (dx, (partials, ntiles))."""
    n, c, _, dsn, dsc = _planes(dy)
    h, w = dy.shape[2] // 2, dy.shape[3] // 2
    dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
    lib = L.lib()
    if bnred is not None and _fuse_bnred:
        a, st = bnred
        _, _, _, asn, asc = _planes(a)
        nt = C.c_int(0)
        check(lib.pcuda_upsample2_bwd_bnred(None, 0, 0, None, 0, 0, 0, None, 0, 0, None, None, None, C.byref(nt), n, c, h, w,
                                            _stream()), "upsample2_bwd_bnred(query)")
        red = torch.empty((nt.value, c, 2), dtype=torch.float32, device=dy.device)
        check(lib.pcuda_upsample2_bwd_bnred(dy.data_ptr(), dsn, dsc, dx.data_ptr(), dx.stride(0), dx.stride(1), 0,
                                            a.data_ptr(), asn, asc, st.mean.data_ptr(), st.invstd.data_ptr(),
                                            red.data_ptr(), C.byref(nt), n, c, h, w, _stream()), "upsample2_bwd_bnred")
        return dx, (red, nt.value)
    check(lib.pcuda_upsample2_bwd(dy.data_ptr(), dsn, dsc, dx.data_ptr(), dx.stride(0), dx.stride(1), 0, n, c, h,
                                  w, _stream()), "upsample2_bwd")
    return (dx, None) if bnred is not None else dx


def bilinear_fwd(x, oh, ow):
    """[n,c,h,w] -> [n,c,oh,ow], align_corners=True (nn.UpsamplingBilinear2d)"""
    n, c, _, xsn, xsc = _planes(x)
    h, w = x.shape[2], x.shape[3]
    y = torch.empty((n, c, oh, ow), dtype=torch.float32, device=x.device)
    check(L.lib().pcuda_bilinear_fwd(x.data_ptr(), xsn, xsc, n, c, h, w, y.data_ptr(), oh, ow, _stream()), "bilinear_fwd")
    return y


def bilinear_bwd(dy, h, w):
    _req(dy)
    dy = dy.contiguous()
    n, c, oh, ow = dy.shape
    dx = torch.empty((n, c, h, w), dtype=torch.float32, device=dy.device)
    check(L.lib().pcuda_bilinear_bwd(dy.data_ptr(), n, c, oh, ow, dx.data_ptr(), dx.stride(0), dx.stride(1), h, w,
                                     _stream()), "bilinear_bwd")
    return dx


def d1_forward_direct(op, n, h, w):
    """True where ``op.forward`` runs on the direct first-layer kernel (pcuda_conv2d_d1_forward_ok): no unfolded tensor"""
    g = op.geom(n, h, w)
    return bool(L.lib().pcuda_conv2d_d1_forward_ok(C.byref(g)))


def unfold_taps(x, k, stride, pad, dil=1):
    """[n,c,h,w] -> [n,c*k*k,oh,ow]: every tap of a k x k window as its own channel (zero outside the image)"""
    n, c, _, xsn, xsc = _planes(x)
    h, w = x.shape[2], x.shape[3]
    oh = (h + 2 * pad - dil * (k - 1) - 1) // stride + 1
    ow = (w + 2 * pad - dil * (k - 1) - 1) // stride + 1
    u = torch.empty((n, c * k * k, oh, ow), dtype=torch.float32, device=x.device)
    check(L.lib().pcuda_unfold_taps(x.data_ptr(), xsn, xsc, n, c, h, w, k, stride, pad, dil, u.data_ptr(), oh, ow,
                                    _stream()), "unfold_taps")
    return u


def add_n(ts, out=None):
    ts = [t for t in ts if t is not None]
    for t in ts:
        _req(t)
        if not t.is_contiguous():
            raise ValueError("add_n needs contiguous tensors")
    if out is None:
        out = torch.empty_like(ts[0])
    p = [t.data_ptr() for t in ts] + [None] * (4 - len(ts))
    check(L.lib().pcuda_add4(p[0], p[1], p[2], p[3], out.data_ptr(), out.numel(), _stream()), "add4")
    return out


def mul(a, b):
    _req(a); _req(b)
    a, b = a.contiguous(), b.contiguous()
    y = torch.empty_like(a)
    check(L.lib().pcuda_mul(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()), "mul")
    return y


# ------------------------------------------------------------------------------------------
# losses / entropy
# ------------------------------------------------------------------------------------------
def _mode(m):
    return ACT_SIGMOID if m in (ACT_SIGMOID, "sigmoid") else ACT_SOFTMAX


def entropy_fwd(logits, mode="sigmoid", norm=1.0, want_prob=False):
    _req(logits)
    logits = logits.contiguous()
    n, c = logits.shape[:2]
    hw = logits.numel() // (n * c)
    ent = torch.empty_like(logits)
    prob = torch.empty_like(logits) if want_prob else None
    check(L.lib().pcuda_entropy_fwd(logits.data_ptr(), _mode(mode), float(norm), ent.data_ptr(), _ptr(prob), n, c, hw,
                                    _stream()), "entropy_fwd")
    return ent, prob


def entropy_bwd(logits, mode, norm, dent=None, dprob=None, out=None, accumulate=False, dmean=None):
    """``dmean``: device scalar, gradient of the map's mean over batch and pixels (sum over channels)"""
    n, c = logits.shape[:2]
    hw = logits.numel() // (n * c)
    if out is None:
        out = torch.empty_like(logits)
    check(L.lib().pcuda_entropy_bwd2(logits.data_ptr(), _mode(mode), float(norm),
                                     _ptr(None if dent is None else dent.contiguous()),
                                     _ptr(None if dprob is None else dprob.contiguous()),
                                     _ptr(None if dmean is None else dmean.contiguous()), out.data_ptr(),
                                     1 if accumulate else 0, n, c, hw, _stream()), "entropy_bwd")
    return out


def sum_all(x, scale=1.0):
    """scale * sum(x) as a 0-dim device tensor (fixed summation order)"""
    _req(x)
    x = x.contiguous()
    lib = L.lib()
    nb = lib.pcuda_sum_all_workspace_size()
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    out = torch.empty((), dtype=torch.float32, device=x.device)
    check(lib.pcuda_sum_all(x.data_ptr(), x.numel(), float(scale), out.data_ptr(), ws.data_ptr(), nb, _stream()), "sum_all")
    return out


def seg_loss_fwd(logits, onehot, mode="sigmoid"):
    _req(logits)
    _req(onehot, torch.uint8)
    n, c = logits.shape[:2]
    hw = logits.numel() // (n * c)
    lib = L.lib()
    nb = lib.pcuda_seg_loss_workspace_size(n, c, hw)
    ws = torch.empty(nb, dtype=torch.uint8, device=logits.device)
    out2 = torch.empty(2, dtype=torch.float32, device=logits.device)
    check(lib.pcuda_seg_loss_fwd(logits.data_ptr(), onehot.data_ptr(), _mode(mode), n, c, hw, out2.data_ptr(),
                                 ws.data_ptr(), nb, _stream()), "seg_loss_fwd")
    return out2, ws


def seg_loss_bwd(logits, onehot, mode, ws, g_main=None, g_jac=None):
    n, c = logits.shape[:2]
    hw = logits.numel() // (n * c)
    d = torch.empty_like(logits)
    check(L.lib().pcuda_seg_loss_bwd(logits.data_ptr(), onehot.data_ptr(), _mode(mode), n, c, hw, _ptr(g_main),
                                     _ptr(g_jac), d.data_ptr(), ws.data_ptr(), _stream()), "seg_loss_bwd")
    return d


def jaccard_fwd(probs, truth, eps):
    """probs fp32 [n,c,...]; truth one-hot, fp32 or uint8, same shape -> (loss scalar, workspace)"""
    _req(probs)
    if truth.dtype not in (torch.float32, torch.uint8):
        truth = truth.float()      # the reference casts any dtype: ``true.type(probas.type())`` (loss.py:27)
    _req(truth, truth.dtype)
    if truth.shape != probs.shape:
        raise ValueError("jaccard: `true` %r and probabilities %r differ in shape" % (tuple(truth.shape), tuple(probs.shape)))
    probs, truth = probs.contiguous(), truth.contiguous()
    n, c = probs.shape[:2]
    hw = probs.numel() // (n * c)
    lib = L.lib()
    nb = lib.pcuda_jaccard_workspace_size(c)
    ws = torch.empty(nb, dtype=torch.uint8, device=probs.device)
    loss = torch.empty((), dtype=torch.float32, device=probs.device)
    check(lib.pcuda_jaccard_fwd(probs.data_ptr(), truth.data_ptr(), 1 if truth.dtype == torch.uint8 else 0, n, c, hw,
                                float(eps), loss.data_ptr(), ws.data_ptr(), nb, _stream()), "jaccard_fwd")
    return loss, ws, truth


def jaccard_bwd(truth, shape, eps, ws, gout):
    n, c = shape[:2]
    hw = 1
    for d in shape[2:]:
        hw *= d
    dp = torch.empty(shape, dtype=torch.float32, device=truth.device)
    check(L.lib().pcuda_jaccard_bwd(truth.data_ptr(), 1 if truth.dtype == torch.uint8 else 0, n, c, hw, float(eps),
                                    _ptr(gout), dp.data_ptr(), ws.data_ptr(), _stream()), "jaccard_bwd")
    return dp


def bce_const_fwd(x, label, want_acc=False):
    _req(x)
    x = x.contiguous()
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    acc = torch.empty((), dtype=torch.float32, device=x.device) if want_acc else None
    check(L.lib().pcuda_bce_const_fwd(x.data_ptr(), x.numel(), float(label), loss.data_ptr(), _ptr(acc), _stream()),
          "bce_const_fwd")
    return loss, acc


def bce_const_bwd(x, label, gout, gscale=1.0):
    x = x.contiguous()
    dx = torch.empty_like(x)
    check(L.lib().pcuda_bce_const_bwd(x.data_ptr(), x.numel(), float(label), _ptr(gout), float(gscale), dx.data_ptr(),
                                      _stream()), "bce_const_bwd")
    return dx


def nn_loss_fwd(x, y):
    _req(x); _req(y)
    x, y = x.contiguous(), y.contiguous()
    b, npts = x.shape[0], x.shape[1]
    idx = torch.empty((2, b, npts), dtype=torch.int32, device=x.device)
    val = torch.empty(L.lib().pcuda_nn_loss_workspace_floats(b, npts), dtype=torch.float32, device=x.device)
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    check(L.lib().pcuda_nn_loss_fwd(x.data_ptr(), y.data_ptr(), b, npts, loss.data_ptr(), idx.data_ptr(),
                                    val.data_ptr(), _stream()), "nn_loss_fwd")
    return loss, idx, val


def nn_loss_bwd(x, y, idx, val, gout):
    x, y = x.contiguous(), y.contiguous()
    b, npts = x.shape[0], x.shape[1]
    dx = torch.empty_like(x)
    check(L.lib().pcuda_nn_loss_bwd(x.data_ptr(), y.data_ptr(), b, npts, idx.data_ptr(), val.data_ptr(), _ptr(gout),
                                    dx.data_ptr(), _stream()), "nn_loss_bwd")
    return dx


def dice_metric(logits, onehot):
    _req(logits); _req(onehot, torch.uint8)
    logits = logits.contiguous()
    n, c = logits.shape[:2]
    hw = logits.numel() // (n * c)
    ws = torch.empty(3 * c, dtype=torch.int64, device=logits.device)
    out = torch.empty((), dtype=torch.float32, device=logits.device)
    check(L.lib().pcuda_dice_metric(logits.data_ptr(), onehot.data_ptr(), n, c, hw, out.data_ptr(), ws.data_ptr(),
                                    ws.numel() * 8, _stream()), "dice_metric")
    return out


def assemble_batch(images_hwc, mask_labels, num_classes, crop=0):
    """[B,H,W,C] fp32 images (+ [B,H,W] int32 labels) -> ([B,C,h,w] fp32, [B,K,h,w] uint8 one-hot | None)."""
    _req(images_hwc)
    images_hwc = images_hwc.contiguous()
    b, h, w, c = images_hwc.shape
    oh, ow = (2 * (crop // 2), 2 * (crop // 2)) if crop else (h, w)
    out = torch.empty((b, c, oh, ow), dtype=torch.float32, device=images_hwc.device)
    onehot = None
    if mask_labels is not None:
        _req(mask_labels, torch.int32)
        mask_labels = mask_labels.contiguous()
        if tuple(mask_labels.shape) != (b, h, w):
            raise ValueError("assemble_batch: mask shape %r does not match the images" % (tuple(mask_labels.shape),))
        onehot = torch.empty((b, num_classes, oh, ow), dtype=torch.uint8, device=images_hwc.device)
    check(L.lib().pcuda_assemble_batch(images_hwc.data_ptr(), _ptr(mask_labels), b, h, w, c, int(crop), int(num_classes),
                                       out.data_ptr(), _ptr(onehot), _stream()), "assemble_batch")
    return out, onehot


def argmax_labels(x):
    """[N,C,H,W] fp32 logits or uint8 one-hot -> uint8 label map [N,H,W]: first channel holding the maximum."""
    if x.dtype not in (torch.float32, torch.uint8):
        raise TypeError("argmax_labels: fp32 logits or a uint8 one-hot mask")
    _req(x, x.dtype)
    n, c = x.shape[:2]
    hw = x.numel() // (n * c)
    if x.stride(-1) != 1 or (x.dim() == 4 and x.stride(2) != x.shape[3]):
        x = x.contiguous()
    lab = torch.empty((n,) + tuple(x.shape[2:]), dtype=torch.uint8, device=x.device)
    check(L.lib().pcuda_argmax_labels(x.data_ptr(), 1 if x.dtype == torch.uint8 else 0, x.stride(0), x.stride(1), n, c,
                                      hw, lab.data_ptr(), _stream()), "argmax_labels")
    return lab


def label_dice(pred_labels, gt_labels, num_classes):
    """per-class Dice (medpy dc) of two uint8 label maps -> fp32 [num_classes] on the device"""
    _req(pred_labels, torch.uint8); _req(gt_labels, torch.uint8)
    if pred_labels.shape != gt_labels.shape:
        raise ValueError("label_dice: shapes differ")
    pred_labels, gt_labels = pred_labels.contiguous(), gt_labels.contiguous()
    ws = torch.empty(3 * num_classes, dtype=torch.int64, device=pred_labels.device)
    out = torch.empty(num_classes, dtype=torch.float32, device=pred_labels.device)
    check(L.lib().pcuda_label_dice(pred_labels.data_ptr(), gt_labels.data_ptr(), pred_labels.numel(), num_classes,
                                   out.data_ptr(), ws.data_ptr(), ws.numel() * 8, _stream()), "label_dice")
    return out


# ------------------------------------------------------------------------------------------
# dense
# ------------------------------------------------------------------------------------------
def linear_fwd(x, w, b):
    _req(x); _req(w)
    m, k = x.shape
    n = w.shape[0]
    y = torch.empty((m, n), dtype=torch.float32, device=x.device)
    check(L.lib().pcuda_linear_fwd(x.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), m, k, n, _stream()), "linear_fwd")
    return y


def linear_bwd_x(dy, w):
    m, n = dy.shape
    k = w.shape[1]
    dx = torch.empty((m, k), dtype=torch.float32, device=dy.device)
    check(L.lib().pcuda_linear_bwd_x(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), m, k, n, 0, _stream()), "linear_bwd_x")
    return dx


def linear_bwd_w(dy, x, dw, db, accumulate=True):
    m, n = dy.shape
    k = x.shape[1]
    lib = L.lib()
    nb = lib.pcuda_linear_bwd_w_workspace_size(m, k, n)
    ws = torch.empty(nb, dtype=torch.uint8, device=dy.device) if nb else None
    check(lib.pcuda_linear_bwd_w(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), _ptr(db), m, k, n,
                                 1 if accumulate else 0, _ptr(ws), nb, _stream()), "linear_bwd_w")


def conv1d_fwd(x, w, b, want_stats=False):
    """torch.nn.Conv1d(cin, cout, 1) on [B,cin,L] in exact fp32 (MFMA f32): -> (y [B,cout,L], bn partials|None, ntiles)"""
    _req(x); _req(w)
    if not (x.is_contiguous() and w.is_contiguous()):
        raise ValueError("conv1d_fwd needs contiguous tensors")
    bsz, cin, l = x.shape
    cout = w.shape[0]
    lib = L.lib()
    y = torch.empty((bsz, cout, l), dtype=torch.float32, device=x.device)
    part, nt = None, 0
    if want_stats:
        nt = lib.pcuda_conv1d_k1_fwd_tiles(bsz, l)
        part = torch.empty((nt, cout, 2), dtype=torch.float32, device=x.device)
    check(lib.pcuda_conv1d_k1_fwd(x.data_ptr(), w.data_ptr(), _ptr(b), y.data_ptr(), bsz, cin, cout, l, _ptr(part),
                                  _stream()), "conv1d_k1_fwd")
    return y, part, nt


def conv1d_dgrad(dy, w):
    _req(dy); _req(w)
    dy = dy.contiguous()
    bsz, cout, l = dy.shape
    cin = w.shape[1]
    dx = torch.empty((bsz, cin, l), dtype=torch.float32, device=dy.device)
    check(L.lib().pcuda_conv1d_k1_dgrad(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), bsz, cin, cout, l, _stream()),
          "conv1d_k1_dgrad")
    return dx


def conv1d_wgrad(x, dy, dw, db, accumulate=True):
    _req(x); _req(dy)
    dy = dy.contiguous()
    bsz, cin, l = x.shape
    cout = dy.shape[1]
    lib = L.lib()
    nb = lib.pcuda_conv1d_k1_wgrad_workspace_size(bsz, cin, cout, l)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    check(lib.pcuda_conv1d_k1_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), _ptr(db), bsz, cin, cout, l,
                                    1 if accumulate else 0, ws.data_ptr(), nb, _stream()), "conv1d_k1_wgrad")


def max_points_fwd(x):
    _req(x)
    b, c, l = x.shape
    y = torch.empty((b, c), dtype=torch.float32, device=x.device)
    idx = torch.empty((b, c), dtype=torch.int32, device=x.device)
    check(L.lib().pcuda_max_points_fwd(x.data_ptr(), b, c, l, y.data_ptr(), idx.data_ptr(), _stream()), "max_points_fwd")
    return y, idx


def max_points_bwd(dy, idx, l):
    b, c = dy.shape
    dx = torch.empty((b, c, l), dtype=torch.float32, device=dy.device)
    check(L.lib().pcuda_max_points_bwd(dy.data_ptr(), idx.data_ptr(), b, c, l, dx.data_ptr(), _stream()), "max_points_bwd")
    return dx


def bmm(a, b, ta=False, tb=False):
    """C[i] = op(A[i]) op(B[i]); a: [B,m,k] (or [B,k,m] with ta), b: [B,k,n] (or [B,n,k] with tb)."""
    _req(a); _req(b)
    batch = a.shape[0]
    m, k = (a.shape[2], a.shape[1]) if ta else (a.shape[1], a.shape[2])
    n = b.shape[1] if tb else b.shape[2]
    c = torch.empty((batch, m, n), dtype=torch.float32, device=a.device)
    check(L.lib().pcuda_bmm(a.data_ptr(), b.data_ptr(), c.data_ptr(), batch, m, k, n, 1 if ta else 0, 1 if tb else 0, 0,
                            _stream()), "bmm")
    return c


# ------------------------------------------------------------------------------------------
# sampler
# ------------------------------------------------------------------------------------------
def surface_vertices(mask_u8: torch.Tensor, max_verts: int, order: str = "lex"):
    """order "lex": this build's canonical list; "mc": marching-cubes traversal order with coincident duplicates"""
    _req(mask_u8, torch.uint8)
    b, h, w = mask_u8.shape
    verts = torch.zeros((b, max_verts, 3), dtype=torch.int32, device=mask_u8.device)
    counts = torch.empty(b, dtype=torch.int32, device=mask_u8.device)
    if order == "mc":
        check(L.lib().pcuda_surface_vertices_mc(mask_u8.data_ptr(), b, h, w, verts.data_ptr(), max_verts,
                                                counts.data_ptr(), _stream()), "surface_vertices_mc")
        return verts, counts
    if order != "lex":
        raise ValueError("order must be 'lex' or 'mc'")
    check(L.lib().pcuda_surface_vertices(mask_u8.data_ptr(), b, h, w, verts.data_ptr(), max_verts, counts.data_ptr(),
                                         None, 0, _stream()), "surface_vertices")
    return verts, counts


def fps(pts_f64: torch.Tensor, counts: torch.Tensor, first: torch.Tensor, k: int):
    _req(pts_f64, torch.float64); _req(counts, torch.int32); _req(first, torch.int32)
    b, npts_max, _ = pts_f64.shape
    idx = torch.empty((b, k), dtype=torch.int32, device=pts_f64.device)
    check(L.lib().pcuda_fps(pts_f64.data_ptr(), counts.data_ptr(), first.data_ptr(), b, npts_max, k, idx.data_ptr(),
                            _stream()), "fps")
    return idx


# ------------------------------------------------------------------------------------------
# optimiser steps
# ------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    check(L.lib().pcuda_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1, beta2,
                                  eps, weight_decay, step, grad_scale, _stream()), "adam_step")


def adam_step_dev(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step_t, grad_scale=1.0):
    """Adam with the step count in the int32 device tensor ``step_t`` (incremented by the call): graph-capturable."""
    check(L.lib().pcuda_adam_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, beta1,
                                      beta2, eps, weight_decay, step_t.data_ptr(), grad_scale, _stream()),
          "adam_step_dev")


def sgd_step(p, g, mom, lr, momentum, weight_decay, first_step, grad_scale=1.0):
    check(L.lib().pcuda_sgd_step(p.data_ptr(), g.data_ptr(), _ptr(mom), p.numel(), lr, momentum, weight_decay,
                                 1 if first_step else 0, grad_scale, _stream()), "sgd_step")


# ------------------------------------------------------------------------------------------
# profiling
# ------------------------------------------------------------------------------------------
def prof_enable(on: bool):
    check(L.lib().pcuda_prof_enable(1 if on else 0))


def prof_reset():
    check(L.lib().pcuda_prof_reset())


def prof_read(family: int):
    ms, work, n = C.c_double(0), C.c_double(0), C.c_longlong(0)
    check(L.lib().pcuda_prof_read(family, C.byref(ms), C.byref(work), C.byref(n)), "prof_read")
    return ms.value, work.value, n.value


def launch_count(reset: bool = False) -> int:
    """kernel launches issued by the library since the last reset (host-side counter)"""
    return int(L.lib().pcuda_launch_count(1 if reset else 0))


def last_kernel() -> str:
    """"<shape tag> | <kernel>" of this thread's most recent convolution launch (``pcuda_last_kernel``): kernel-level tests
    assert the dispatch with it"""
    v = L.lib().pcuda_last_kernel()
    return v.decode() if v else ""


def clock_ghz_under_load(dev, iters: int = 8192, reps: int = 3) -> float:
    """shader clock (GHz) while every SIMD of the chip issues MFMAs back to back (``pcuda_clock_probe``): the last of ``reps``
    ~3-ms launches, so that the power state has settled"""
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    sink = torch.zeros(1, dtype=torch.float32, device=dev)
    ghz = 0.0
    for _ in range(reps):
        check(L.lib().pcuda_clock_probe(out.data_ptr(), sink.data_ptr(), int(iters), _stream()), "clock_probe")
        c, r = (int(v) for v in out.tolist())
        ghz = 0.1 * c / max(r, 1)
    return ghz


def fallback_count() -> int:
    return int(L.lib().pcuda_fallback_count())


def prof_dump(path: str):
    check(L.lib().pcuda_prof_dump(path.encode()), "prof_dump")
