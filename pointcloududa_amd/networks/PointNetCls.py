"""HIP-backed drop-in for the reference's ``src/networks/PointNetCls.py`` (the d4 discriminator).

Same classes / constructor arguments / parameter names (PointNetCls.py:11-224).  The k=1 Conv1d
layers run in EXACT fp32 on the matrix cores (csrc/conv1d_f32.hip: v_mfma_f32_32x32x2_f32, forward, data and weight
gradient), with the BatchNorm1d partial statistics taken from the forward kernel's epilogue; Linear layers, the
3x3 / 64x64 transforms and the max over points use the small dense kernels.  The whole classifier
is one autograd node with a recorded tape of backward steps; parameter gradients are accumulated
into ``param.grad`` directly.

Batch size 1 takes the reference down an InstanceNorm path (PointNetCls.py:47-55, 210-212) that raises inside
the reference itself (InstanceNorm1d on the 2-D output of fc1): the same RuntimeError is raised here, and per-rank
batches must be >= 2, as SURVEY section 7 notes for data parallelism.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from .. import kernels as K
from ._holders import BatchNorm1d, Conv1d, InstanceNorm1d, Linear, Marker, ensure_grad


class _Var:
    __slots__ = ("t", "g")

    def __init__(self, t):
        self.t, self.g = t, None

    def acc(self, g):
        self.g = g if self.g is None else K.add_n([self.g, g])


class _Tape:
    """forward ops record their backward closure; run() replays them in reverse"""

    def __init__(self, module, training):
        self.m, self.training, self.steps = module, training, []
        self.P = dict(module.named_parameters())
        # which parameters want a gradient is decided when the forward pass is recorded, as autograd does
        self.wants = {k for k, p in self.P.items() if p.requires_grad}
        self.P.update(dict(module.named_buffers()))
        # tests (shared-routing backward checks): name -> intermediate tensor of this forward pass
        self.trace = {} if getattr(module, "_keep_trace", False) else None

    def _rec(self, name, t, relu=None):
        if self.trace is not None:
            self.trace[name] = t if relu is None else (t, relu)

    def G(self, name):
        return ensure_grad(self.P[name]) if name in self.wants else None

    def run(self):
        for fn in reversed(self.steps):
            fn()
        self.steps = []

    # -- ops ---------------------------------------------------------------
    def conv_bn(self, x: _Var, conv: str, bn: str, relu: bool) -> _Var:
        """Conv1d(k=1) -> BatchNorm1d [-> ReLU] on [B,C,L]"""
        w, b = self.P[conv + ".weight"], self.P[conv + ".bias"]
        if w.shape[2] != 1:
            raise NotImplementedError("PointNet Conv1d kernel_size must be 1")
        bsz, cin, l = x.t.shape
        cout = w.shape[0]
        w2 = w.view(cout, cin)
        a_t, part, nt = K.conv1d_fwd(x.t, w2, b, want_stats=self.training)
        a = _Var(a_t)
        self._rec(conv, a.t)
        y = self._bn(a, bn, relu, part, nt)

        def bwd():
            if a.g is None:
                return
            dz = a.g.contiguous()
            if self.G(conv + ".weight") is not None:
                K.conv1d_wgrad(x.t, dz, self.G(conv + ".weight").view(cout, cin), self.G(conv + ".bias"))
            x.acc(K.conv1d_dgrad(dz, w2))
        # the BN backward was recorded after this closure's position would be wrong: insert before it
        self.steps.insert(len(self.steps) - 1, bwd)
        return y

    def _bn(self, a: _Var, bn: str, relu: bool, part=None, nt=0) -> _Var:
        P = self.P
        n, c = a.t.shape[0], a.t.shape[1]
        cnt = a.t.numel() // c
        if self.training:
            if part is None:
                part, nt, cnt = K.bn_stats(a.t)
            st = K.bn_finalize(part, nt, cnt, P[bn + ".weight"], P[bn + ".bias"], P[bn + ".running_mean"],
                               P[bn + ".running_var"])
        else:
            st = K.BNState()
            inv = torch.rsqrt(P[bn + ".running_var"] + 1e-5)
            st.mean, st.invstd, st.count = P[bn + ".running_mean"], inv, cnt
            st.scale = (P[bn + ".weight"] * inv).contiguous()
            st.shift = (P[bn + ".bias"] - P[bn + ".running_mean"] * st.scale).contiguous()
        y = _Var(K.bn_apply(a.t, st, relu=relu))
        self._rec(bn, y.t, relu)

        def bwd():
            if y.g is None:
                return
            a.acc(K.bn_backward(y.g, a.t, st, P[bn + ".weight"], self.G(bn + ".weight"), self.G(bn + ".bias"),
                                post_relu=relu, act_slope=1.0, frozen=not self.training))
        self.steps.append(bwd)
        return y

    def linear(self, x: _Var, name: str, bias_plus=None) -> _Var:
        w, b = self.P[name + ".weight"], self.P[name + ".bias"]
        b_eff = b if bias_plus is None else K.add_n([b, bias_plus])
        y = _Var(K.linear_fwd(x.t, w, b_eff))
        self._rec(name, y.t)

        def bwd():
            if y.g is None:
                return
            if self.G(name + ".weight") is not None:
                K.linear_bwd_w(y.g, x.t, self.G(name + ".weight"), self.G(name + ".bias"))
            x.acc(K.linear_bwd_x(y.g, w))
        self.steps.append(bwd)
        return y

    def linear_bn(self, x: _Var, fc: str, bn: str, mask=None) -> _Var:
        """Linear [-> dropout mask] -> BatchNorm1d -> ReLU on [B,C]"""
        h = self.linear(x, fc)
        if mask is not None:
            h = self.mul_const(h, mask)
        return self._bn(h, bn, True)

    def mul_const(self, x: _Var, mask) -> _Var:
        y = _Var(K.mul(x.t, mask))

        def bwd():
            if y.g is not None:
                x.acc(K.mul(y.g, mask))
        self.steps.append(bwd)
        return y

    def max_points(self, x: _Var) -> _Var:
        v, idx = K.max_points_fwd(x.t)
        y = _Var(v)
        l = x.t.shape[2]

        def bwd():
            if y.g is not None:
                x.acc(K.max_points_bwd(y.g.contiguous(), idx, l))
        self.steps.append(bwd)
        return y

    def transform(self, x: _Var, trans: _Var) -> _Var:
        """x'[b] = trans[b]^T . x[b]   (= bmm(x^T, trans)^T, PointNetCls.py:140-142,148-151)"""
        y = _Var(K.bmm(trans.t, x.t, ta=True))

        def bwd():
            if y.g is None:
                return
            g = y.g.contiguous()
            x.acc(K.bmm(trans.t, g))                 # dX = T . dY
            trans.acc(K.bmm(x.t, g, tb=True))        # dT = X . dY^T
        self.steps.append(bwd)
        return y

    def stn(self, x: _Var, pre: str, k: int) -> _Var:
        h = self.conv_bn(x, pre + "conv1", pre + "bn1", True)
        h = self.conv_bn(h, pre + "conv2", pre + "bn2", True)
        h = self.conv_bn(h, pre + "conv3", pre + "bn3", True)
        g = self.max_points(h)
        g = self.linear_bn(g, pre + "fc1", pre + "bn4")
        g = self.linear_bn(g, pre + "fc2", pre + "bn5")
        iden = self.m._identity(k, x.t.device)
        t = self.linear(g, pre + "fc3", bias_plus=iden)
        out = _Var(t.t.view(-1, k, k))

        def bwd():
            if out.g is not None:
                t.acc(out.g.contiguous().view(-1, k * k))
        self.steps.append(bwd)
        return out


class _PointNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, drop_mask, *params):
        if x.shape[0] < 2:
            # The reference's batch-1 branch (PointNetCls.py:47-55) hands the 2-D [1, 512] output of fc1 to
            # InstanceNorm1d(512): torch 1.4 (the reference's version) raises "InstanceNorm1d returns 0-filled tensor
            # to 2D tensor", current torch raises "running_mean should contain 1 elements not 512" -- the branch cannot
            # execute in the reference either (pinned in tests/golden/param_counts.npz: pncls_batch1_raises).  Same
            # error behaviour here: a RuntimeError naming the layer.
            raise RuntimeError("PointNetCls: batch size 1 takes the reference's InstanceNorm1d branch, which fails in "
                               "the reference too (feat.stn.in4 on a 2-D [1, 512] tensor: running_mean should contain "
                               "1 elements not 512); use a per-rank batch of at least 2")
        if not x.is_cuda:
            raise RuntimeError("PointNetCls runs on HIP devices only (no CPU fallback)")
        tape = _Tape(module, module.training)
        xin = _Var(x.contiguous().float())
        y, trans, trans_feat = module._run(tape, xin, drop_mask)
        if tape.trace is not None:
            module._last_trace = tape.trace
        if module.training:
            flat = getattr(module, "_flat_tracked", None)
            if flat is not None:   # (still the buffers' storage?  module.to() / a re-registered buffer leaves the flat tensor behind)
                first = next((b for k, b in module.named_buffers() if k.endswith("num_batches_tracked")), None)
                if first is None or first.data_ptr() != flat.data_ptr():
                    flat = None
            if flat is not None:
                # (optim.flatten_module re-seated every num_batches_tracked buffer as a view of ONE int64 tensor, in
                #  named_buffers() order: one launch with a 0 / 1 mask instead of one per BatchNorm layer -- 32 per step)
                mask = getattr(module, "_tracked_mask", None)
                if mask is None or mask.device != flat.device or mask.numel() != flat.numel():
                    names = [k for k, _ in module.named_buffers() if k.endswith("num_batches_tracked")]
                    mask = torch.tensor([1 if (".in" not in k and not k.startswith("in")) else 0 for k in names],
                                        dtype=torch.long, device=flat.device)
                    module._tracked_mask = mask
                flat.add_(mask)
            else:
                for k, b in module.named_buffers():
                    if k.endswith("num_batches_tracked") and ".in" not in k and not k.startswith("in"):
                        b.add_(1)
        ctx.tape, ctx.vars = tape, (xin, y, trans, trans_feat)
        ctx.set_materialize_grads(False)
        outs = (y.t, trans.t if trans is not None else x.new_zeros(()),
                trans_feat.t if trans_feat is not None else x.new_zeros(()))
        ctx.mark_non_differentiable(*[o for o, v in zip(outs[1:], (trans, trans_feat)) if v is None])
        return outs

    @staticmethod
    def backward(ctx, dy, dtrans, dtrans_feat):
        xin, y, trans, trans_feat = ctx.vars
        if dy is not None:
            y.acc(dy.contiguous())
        if dtrans is not None and trans is not None:
            trans.acc(dtrans.contiguous())
        if dtrans_feat is not None and trans_feat is not None:
            trans_feat.acc(dtrans_feat.contiguous())
        ctx.tape.run()
        dx = xin.g if ctx.needs_input_grad[1] else None
        ctx.tape = ctx.vars = None
        return (None, dx, None) + (None,) * (len(ctx.needs_input_grad) - 3)


# ============================================================================ module tree (reference names)
def _stn_members(mod, cin, kout, with_in):
    mod.conv1 = Conv1d(cin, 64, 1)
    mod.conv2 = Conv1d(64, 128, 1)
    mod.conv3 = Conv1d(128, 1024, 1)
    mod.fc1 = Linear(1024, 512)
    mod.fc2 = Linear(512, 256)
    mod.fc3 = Linear(256, kout)
    mod.relu = Marker("ReLU")
    mod.bn1, mod.bn2, mod.bn3 = BatchNorm1d(64), BatchNorm1d(128), BatchNorm1d(1024)
    mod.bn4, mod.bn5 = BatchNorm1d(512), BatchNorm1d(256)
    if with_in:
        mod.in1 = InstanceNorm1d(64, track_running_stats=True)
        mod.in2 = InstanceNorm1d(128, track_running_stats=True)
        mod.in3 = InstanceNorm1d(1024, track_running_stats=True)
        mod.in4 = InstanceNorm1d(512, track_running_stats=True)
        mod.in5 = InstanceNorm1d(256, track_running_stats=True)


class STN3d(nn.Module):
    def __init__(self, dim=3):
        super().__init__()
        _stn_members(self, dim, 9, with_in=True)


class STNkd(nn.Module):
    def __init__(self, k=64):
        super().__init__()
        _stn_members(self, k, k * k, with_in=False)
        self.k = k


class PointNetfeat(nn.Module):
    def __init__(self, global_feat=True, feature_transform=False, sample_transform=True, kernel_size=1, stride=1,
                 in_channel=3, dim=3, ext=False):
        super().__init__()
        if kernel_size != 1 or stride != 1:
            raise NotImplementedError("PointNetfeat: only kernel_size=1, stride=1 (the reference's defaults)")
        if not global_feat:
            raise NotImplementedError("PointNetfeat(global_feat=False) is never used by PointNetCls")
        self.stn = STN3d(dim=dim)
        self._ext = ext
        if ext:
            self.conv1, self.bn1 = Conv1d(in_channel, 8, 1), BatchNorm1d(8)
            self.conv1_1, self.bn1_1 = Conv1d(8, 64, 1), BatchNorm1d(64)
            self.conv2, self.bn2 = Conv1d(64, 128, 1), BatchNorm1d(128)
            self.conv2_1, self.bn2_1 = Conv1d(128, 256, 1), BatchNorm1d(256)
            self.conv3, self.bn3 = Conv1d(256, 512, 1), BatchNorm1d(512)
            self.conv3_1, self.bn3_1 = Conv1d(512, 1024, 1), BatchNorm1d(1024)
        else:
            self.conv1 = Conv1d(in_channel, 64, 1)
            self.conv2 = Conv1d(64, 128, 1)
            self.conv3 = Conv1d(128, 1024, 1)
            self.bn1, self.bn2, self.bn3 = BatchNorm1d(64), BatchNorm1d(128), BatchNorm1d(1024)
        self.global_feat = global_feat
        self.feature_transform = feature_transform
        self._sample_transform = sample_transform
        if feature_transform:
            self.fstn = STNkd(k=64)


class PointNetCls(nn.Module):
    """PointNetCls.py:170-214.  forward(x[B,3,N]) -> (logit[B,1], trans[B,3,3]|None, trans_feat|None).

    ``drop_mask`` (optional, [B,256] with entries 0 or 1/(1-p)) replaces the nn.Dropout draw for
    deterministic parity runs; by default a mask is drawn from torch's RNG when training with p > 0.
    """

    def __init__(self, feature_transform=False, sample_transform=True, kernel_size=1, stride=1, in_channel=3, dim=3,
                 ext=False, drop=0.3, heinit=False, cvinit=False):
        super().__init__()
        self.feature_transform = feature_transform
        self.feat = PointNetfeat(global_feat=True, feature_transform=feature_transform,
                                 sample_transform=sample_transform, kernel_size=kernel_size, stride=stride,
                                 in_channel=in_channel, dim=dim, ext=ext)
        self.fc1, self.fc2, self.fc3 = Linear(1024, 512), Linear(512, 256), Linear(256, 1)
        self.dropout = Marker("Dropout(p=%g)" % drop)
        self._p = float(drop)
        self.bn1, self.bn2 = BatchNorm1d(512), BatchNorm1d(256)
        self.in1 = InstanceNorm1d(512, track_running_stats=True)
        self.in2 = InstanceNorm1d(256, track_running_stats=True)
        self.relu = Marker("ReLU")
        # heinit / cvinit only touch nn.Conv2d modules in the reference (PointNetCls.py:187-202): none exist here
        self._eyes = {}

    def _identity(self, k, device):
        key = (k, str(device))
        if key not in self._eyes:
            self._eyes[key] = torch.eye(k, dtype=torch.float32, device=device).reshape(k * k).contiguous()
        return self._eyes[key]

    def _run(self, tape: _Tape, x: _Var, drop_mask):
        f = self.feat
        trans = trans_feat = None
        if f._sample_transform:
            trans = tape.stn(x, "feat.stn.", 3)
            x = tape.transform(x, trans)
        h = tape.conv_bn(x, "feat.conv1", "feat.bn1", True)
        if f._ext:
            h = tape.conv_bn(h, "feat.conv1_1", "feat.bn1_1", True)
        if f.feature_transform:
            trans_feat = tape.stn(h, "feat.fstn.", 64)
            h = tape.transform(h, trans_feat)
        h = tape.conv_bn(h, "feat.conv2", "feat.bn2", True)
        if f._ext:
            h = tape.conv_bn(h, "feat.conv2_1", "feat.bn2_1", True)
        h = tape.conv_bn(h, "feat.conv3", "feat.bn3", False)
        if f._ext:
            h = tape.conv_bn(h, "feat.conv3_1", "feat.bn3_1", True)
        g = tape.max_points(h)
        g = tape.linear_bn(g, "fc1", "bn1")
        g = tape.linear_bn(g, "fc2", "bn2", mask=drop_mask)
        y = tape.linear(g, "fc3")
        return y, trans, trans_feat

    def forward(self, x, drop_mask=None):
        if drop_mask is None and self.training and self._p > 0:
            keep = 1.0 - self._p
            drop_mask = torch.bernoulli(torch.full((x.shape[0], 256), keep, device=x.device)) / keep
        params = list(self.parameters())
        y, trans, trans_feat = _PointNetFn.apply(self, x, drop_mask, *params)
        return (y, trans if self.feat._sample_transform else None,
                trans_feat if self.feature_transform else None)


def feature_transform_regularizer(trans):
    """PointNetCls.py:217-224 (unused by the train scripts): mean ||T T^T - I||_F.  Launch-bound
    scalar bookkeeping on a [B,d,d] tensor; kept on torch ops."""
    d = trans.size()[1]
    eye = torch.eye(d, device=trans.device)[None, :, :]
    return torch.mean(torch.norm(torch.bmm(trans, trans.transpose(2, 1)) - eye, dim=(1, 2)))
