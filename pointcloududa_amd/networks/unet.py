"""HIP-backed drop-in for the reference's ``src/networks/unet.py``.

Same classes, constructor signatures, module tree and ``state_dict`` keys as the reference
(unet.py:7-233); the arithmetic is one fused autograd node per network call that launches the
libpcuda_hip kernels (implicit-GEMM MFMA convolutions with fused bias + LeakyReLU + BatchNorm
partial statistics, BatchNorm applied lazily inside the consumer's load, nearest-upsample and
channel concatenation folded into the convolution's addressing).

Gradients of the parameters are accumulated straight into ``param.grad`` (created on first use)
instead of being returned through autograd: the train step keeps them in one flat buffer per
network, which is also what gets all-reduced over RCCL.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from .. import kernels as K
from ..kernels import TA, ConvOp
from ._holders import (BatchNorm2d, Conv2d, LeakyReLU, Linear, Marker, collect, ensure_grad)

SLOPE = 0.01   # nn.LeakyReLU() default used throughout unet.py


# ============================================================================ engine


class _SegEngine:
    """forward / hand-written backward of Segmentation_model_Point over a name->tensor dict"""

    def __init__(self, filters, in_channels, n_block, depth, n_class, pointnet, fc_inch, extpn=False, batchnorm=True,
                 feature_dis=False):
        f = filters
        self.f, self.cin, self.nb, self.depth, self.ncls, self.pointnet, self.fc_inch = \
            f, in_channels, n_block, depth, n_class, pointnet, fc_inch
        self.extpn = bool(extpn and pointnet)
        self.bn = bool(batchnorm)
        self.c2 = ".3" if self.bn else ".2"      # index of a block's second convolution inside its nn.Sequential (unet.py:23-30)
        self.feature_dis = bool(feature_dis)
        self.after_deep_grads = None      # optional hook of the backward pass (see ``backward``)
        ops = {}
        for i in range(n_block):
            co = f * 2 ** i
            ci = in_channels if i == 0 else f * 2 ** (i - 1)
            blk = "encoder.encoder%d" % (i + 1)
            ops[blk + ".0"] = ConvOp(ci, co, 3, pad=1)
            ops[blk + self.c2] = ConvOp(co, co, 3, pad=1)
            if i > 0:
                ops["encoder.conv1_%d.0" % (i + 1)] = ConvOp(ci * 3, co, 1)
        co, ci = f * 2 ** n_block, f * 2 ** (n_block - 1)
        for j in range(depth):
            d = 2 ** j
            ops["bottleneck.bottleneck%d.0" % (j + 1)] = ConvOp(ci, co, 3, pad=d, dil=d)
            ci = co
        if pointnet:
            ch = 512 * f // 32
            if self.extpn:                                            # unet.py:81-83
                ops["pointNet.conv1"] = ConvOp(ch, 2 * ch, 3, pad=1)
                ops["pointNet.conv2"] = ConvOp(2 * ch, ch, 3, pad=1)
            ops["pointNet.final_conv"] = ConvOp(ch, 300, 6)
        for i in range(n_block):
            co = f * 2 ** i
            ops["decoder.decoder1_%d.1" % (i + 1)] = ConvOp(2 * co, co, 3, pad=1, in_up=True)
            blk = "decoder.decoder2_%d" % (i + 1)
            ops[blk + ".0"] = ConvOp(2 * co, co, 3, pad=1)
            ops[blk + self.c2] = ConvOp(co, co, 3, pad=1)
        ops["classifier"] = ConvOp(f, n_class, 1)
        if self.feature_dis:                                          # unet.py:147-148: hard-wired 512 input channels
            ops["classifier2"] = ConvOp(512, n_class, 1)
        self.ops = ops

    # ---------------------------------------------------------------- helpers
    @staticmethod
    def _bn(P, name, partials, nt, count, training, c_dev):
        if training:
            return K.bn_finalize(partials, nt, count, P[name + ".weight"], P[name + ".bias"],
                                 P[name + ".running_mean"], P[name + ".running_var"])
        st = K.BNState()   # eval: fold the running statistics (tiny [C] vectors; not on the train path)
        inv = torch.rsqrt(P[name + ".running_var"] + 1e-5)
        st.mean, st.invstd = P[name + ".running_mean"], inv
        st.scale = (P[name + ".weight"] * inv).contiguous()
        st.shift = (P[name + ".bias"] - P[name + ".running_mean"] * st.scale).contiguous()
        st.count = count
        return st

    def _dc_fwd(self, P, blk, x, x2, h, w, training, S):
        """conv3x3 -> LeakyReLU -> BN -> conv3x3 -> LeakyReLU -> BN (unet.py:23-30,116-125)."""
        n = (x.t if isinstance(x, TA) else x).shape[0]
        if not self.bn:      # batchnorm=False (unet.py:25,29): conv -> LeakyReLU -> conv -> LeakyReLU
            a0, _, _ = self.ops[blk + ".0"].forward(x, P[blk + ".0.weight"], P[blk + ".0.bias"], SLOPE, h, w, x2=x2)
            a1, _, _ = self.ops[blk + ".2"].forward(a0, P[blk + ".2.weight"], P[blk + ".2.bias"], SLOPE, h, w)
            S[blk] = (x, x2, a0, None, a1, None)
            return a1
        a0, part, nt = self.ops[blk + ".0"].forward(x, P[blk + ".0.weight"], P[blk + ".0.bias"], SLOPE, h, w, x2=x2,
                                                    want_stats=training)
        st0 = self._bn(P, blk + ".2", part, nt, n * h * w, training, a0.device)
        a1, part, nt = self.ops[blk + ".3"].forward(TA(a0, st0.scale, st0.shift), P[blk + ".3.weight"],
                                                    P[blk + ".3.bias"], SLOPE, h, w, want_stats=training)
        st1 = self._bn(P, blk + ".5", part, nt, n * h * w, training, a1.device)
        S[blk] = (x, x2, a0, st0, a1, st1)
        return TA(a1, st1.scale, st1.shift)

    def _dc_bwd(self, P, G, blk, dy, dy2, h, w, S, need_dx, red=None, pooled=None):
        """``red``: the second BatchNorm's backward-reduce partials where the kernel that produced ``dy`` already
        computed them (decoder blocks: the classifier's dgrad, the 2x2 fold behind an up-convolution).
        ``pooled=(g, g2, idx)`` instead of ``dy``: the gradient arrives through the max-pool behind this block
        (unet.py:48) and the first backward kernels read its scatter in place (encoder block 1)"""
        x, x2, a0, st0, a1, st1 = S[blk]
        if not self.bn:
            if pooled is not None:
                dz1 = K.lrelu_bwd_pooled(pooled[0], pooled[2], a1, SLOPE, g2=pooled[1], dy=dy2)
            else:
                dz1 = K.lrelu_bwd(dy, a1, SLOPE, dy2=dy2)
            if G(blk + ".2.weight") is not None:
                self.ops[blk + ".2"].wgrad(a0, dz1, G(blk + ".2.weight"), G(blk + ".2.bias"), h, w)
            dz0 = K.lrelu_bwd(self.ops[blk + ".2"].dgrad(dz1, P[blk + ".2.weight"], h, w), a0, SLOPE)
            return self._dc_bwd_first(P, G, blk, x, x2, dz0, h, w, need_dx)
        frozen = not S["training"]      # eval-mode BatchNorm (running statistics): a fixed affine in the backward pass
        if pooled is not None:
            dz1 = K.bn_backward_pooled(pooled[0], pooled[2], a1, st1, P[blk + ".5.weight"], G(blk + ".5.weight"),
                                       G(blk + ".5.bias"), g2=pooled[1], dy=dy2, act_slope=SLOPE, frozen=frozen)
        else:
            dz1 = K.bn_backward(dy, a1, st1, P[blk + ".5.weight"], G(blk + ".5.weight"), G(blk + ".5.bias"), dy2=dy2,
                                act_slope=SLOPE, red=red if dy2 is None else None, frozen=frozen)
        if G(blk + ".3.weight") is not None:
            self.ops[blk + ".3"].wgrad(TA(a0, st0.scale, st0.shift), dz1, G(blk + ".3.weight"), G(blk + ".3.bias"), h, w)
        # the second convolution's data gradient IS the first BatchNorm's incoming gradient: its reduce (sum g,
        # sum g * a_hat) rides in the dgrad kernel's epilogue where the geometry allows
        d_y0, red = self.ops[blk + ".3"].dgrad(dz1, P[blk + ".3.weight"], h, w, bnred=(a0, st0))
        dz0 = K.bn_backward(d_y0, a0, st0, P[blk + ".2.weight"], G(blk + ".2.weight"), G(blk + ".2.bias"),
                            act_slope=SLOPE, red=red, frozen=frozen)
        return self._dc_bwd_first(P, G, blk, x, x2, dz0, h, w, need_dx)

    def _dc_bwd_first(self, P, G, blk, x, x2, dz0, h, w, need_dx):
        """weight and input gradients of a block's first convolution"""
        if G(blk + ".0.weight") is not None:
            self.ops[blk + ".0"].wgrad(x, dz0, G(blk + ".0.weight"), G(blk + ".0.bias"), h, w, x2=x2)
        if not need_dx:
            return None, None
        op0 = self.ops[blk + ".0"]
        n = dz0.shape[0]
        if x2 is None:
            return op0.dgrad(dz0, P[blk + ".0.weight"], h, w), None
        c1 = (x.t if isinstance(x, TA) else x).shape[1]
        dx1 = torch.empty((n, c1, h, w), dtype=torch.float32, device=dz0.device)
        dx2 = torch.empty((n, op0.cin - c1, h, w), dtype=torch.float32, device=dz0.device)
        op0.dgrad(dz0, P[blk + ".0.weight"], h, w, dx=dx1, dx2=dx2)
        return dx1, dx2

    # ---------------------------------------------------------------- forward
    def forward(self, P, x, training):
        n, _, H, W = x.shape
        nb = self.nb
        if H % (1 << nb) or W % (1 << nb):
            raise ValueError("input size must be divisible by %d" % (1 << nb))
        S = {"hw": (H, W), "n": n, "wants": {k for k, p in P.items() if p.requires_grad}, "training": training}
        cur, h, w, res = x, H, W, None
        skips = []
        for i in range(nb):                                           # unet.py:35-51
            y = self._dc_fwd(P, "encoder.encoder%d" % (i + 1), cur, None, h, w, training, S)
            skips.append(y)
            if i > 0:
                c1 = "encoder.conv1_%d.0" % (i + 1)
                t, _, _ = self.ops[c1].forward(y, P[c1 + ".weight"], P[c1 + ".bias"], SLOPE, h, w, x2=res)
                S[c1] = (y, res, t)
                pooled, idx = K.maxpool2_fwd(t)
            else:
                pooled, idx = K.maxpool2_fwd(y)
            S["pool%d" % i] = idx
            res = cur = pooled
            h, w = h // 2, w // 2
        outs, o = [], cur                                             # unet.py:67-73
        for j in range(self.depth):
            name = "bottleneck.bottleneck%d.0" % (j + 1)
            S[name] = o
            o, _, _ = self.ops[name].forward(o, P[name + ".weight"], P[name + ".bias"], SLOPE, h, w)
            outs.append(o)
        bsum = outs[0]
        for k0 in range(1, len(outs), 3):
            bsum = K.add_n([bsum] + outs[k0:k0 + 3])
        S["bott_outs"] = outs
        verts = None
        if self.pointnet:                                             # unet.py:89-96
            hin, ext_acts = bsum, []
            if self.extpn:                                            # :90-92: two 3x3 convs + LeakyReLU in front
                for nm in ("pointNet.conv1", "pointNet.conv2"):
                    o, _, _ = self.ops[nm].forward(hin, P[nm + ".weight"], P[nm + ".bias"], SLOPE, h, w)
                    ext_acts.append((nm, hin, o))
                    hin = o
            S["head_ext"] = ext_acts
            hc, _, _ = self.ops["pointNet.final_conv"].forward(hin, P["pointNet.final_conv.weight"],
                                                               P["pointNet.final_conv.bias"], SLOPE, h, w)
            flat = hc.view(n * 300, -1)
            if flat.shape[1] != self.fc_inch:
                raise ValueError("fc_inch=%d does not match the %dx%d head output" % (self.fc_inch, h - 5, w - 5))
            verts = K.linear_fwd(flat, P["pointNet.final_fc.weight"], P["pointNet.final_fc.bias"]).view(n, 300, 3)
            S["head"] = (hin, hc, flat)
        out2 = None
        if self.feature_dis:                                          # unet.py:157-158: classifier2(output_bottleneck)
            S["bsum"] = bsum
            out2, _, _ = self.ops["classifier2"].forward(bsum, P["classifier2.weight"], P["classifier2.bias"], 1.0, h, w)
        S["out2"] = out2
        prev, ph, pw = bsum, h, w
        for i in reversed(range(nb)):                                 # unet.py:128-136
            up = "decoder.decoder1_%d.1" % (i + 1)
            oh, ow = 2 * ph, 2 * pw
            S[up] = prev
            u, _, _ = self.ops[up].forward(prev, P[up + ".weight"], P[up + ".bias"], 1.0, oh, ow)
            prev = self._dc_fwd(P, "decoder.decoder2_%d" % (i + 1), skips[i], u, oh, ow, training, S)
            ph, pw = oh, ow
        S["cls_in"] = prev
        logits, _, _ = self.ops["classifier"].forward(prev, P["classifier.weight"], P["classifier.bias"], 1.0, ph, pw)
        return logits, verts, S

    # ---------------------------------------------------------------- backward
    def backward(self, P, S, d_logits, d_verts, need_dx, d_out2=None):
        wants = S["wants"]       # parameters that required a gradient when the forward pass ran (autograd's rule)

        def G(name):
            return ensure_grad(P[name]) if name in wants else None

        nb, (H, W), n = self.nb, S["hw"], S["n"]
        d_skips = [None] * nb
        d_bsum = None
        if d_logits is not None:
            d_logits = d_logits.contiguous()
            if G("classifier.weight") is not None:
                self.ops["classifier"].wgrad(S["cls_in"], d_logits, G("classifier.weight"), G("classifier.bias"), H, W)
            # every decoder block's incoming gradient has ONE producer (the classifier's dgrad, then the 2x2 fold behind
            # each up-convolution): that kernel also computes the block's second BatchNorm's backward-reduce partials
            bn_of = lambda i: (S["decoder.decoder2_%d" % (i + 1)][4], S["decoder.decoder2_%d" % (i + 1)][5])
            if self.bn:
                d_cur, red = self.ops["classifier"].dgrad(d_logits, P["classifier.weight"], H, W, bnred=bn_of(0))
            else:
                d_cur, red = self.ops["classifier"].dgrad(d_logits, P["classifier.weight"], H, W), None
            for i in range(nb):
                oh, ow = H >> i, W >> i
                d_skips[i], d_u = self._dc_bwd(P, G, "decoder.decoder2_%d" % (i + 1), d_cur, None, oh, ow, S, True, red=red)
                up = "decoder.decoder1_%d.1" % (i + 1)
                if G(up + ".weight") is not None:
                    self.ops[up].wgrad(S[up], d_u, G(up + ".weight"), G(up + ".bias"), oh, ow)
                # the up-convolution's data gradient at the STORED resolution: the 2x2 fold of the nearest-x2 backward (and the
                # next block's BatchNorm-backward reduce) ride in the dgrad kernel's epilogue where the plan allows
                if i + 1 < nb and self.bn:
                    d_cur, red = self.ops[up].dgrad_fold(d_u, P[up + ".weight"], oh, ow, bnred=bn_of(i + 1))
                else:
                    d_cur, red = self.ops[up].dgrad_fold(d_u, P[up + ".weight"], oh, ow), None
            d_bsum = d_cur
        h, w = H >> nb, W >> nb
        if self.feature_dis and d_out2 is not None:
            d_out2 = d_out2.contiguous()
            op2 = self.ops["classifier2"]
            if G("classifier2.weight") is not None:
                op2.wgrad(S["bsum"], d_out2, G("classifier2.weight"), G("classifier2.bias"), h, w)
            if d_bsum is None:
                d_bsum = op2.dgrad(d_out2, P["classifier2.weight"], h, w)
            else:
                op2.dgrad(d_out2, P["classifier2.weight"], h, w, dx=d_bsum, accumulate=True)
        if self.pointnet and d_verts is not None:
            hin, hc, flat = S["head"]
            d_v = d_verts.contiguous().view(n * 300, 3)
            if G("pointNet.final_fc.weight") is not None:
                K.linear_bwd_w(d_v, flat, G("pointNet.final_fc.weight"), G("pointNet.final_fc.bias"))
            d_hc = K.linear_bwd_x(d_v, P["pointNet.final_fc.weight"]).view(hc.shape)
            dzc = K.lrelu_bwd(d_hc, hc, SLOPE)
            op, op_name = self.ops["pointNet.final_conv"], "pointNet.final_conv"
            if G("pointNet.final_conv.weight") is not None:
                op.wgrad(hin, dzc, G("pointNet.final_conv.weight"), G("pointNet.final_conv.bias"), h, w)
            for nm, xin, o in reversed(S["head_ext"]):                # extpn: back through conv2, conv1
                d_o = op.dgrad(dzc, P[op_name + ".weight"], h, w)
                dzc = K.lrelu_bwd(d_o, o, SLOPE)
                op, op_name = self.ops[nm], nm
                if G(nm + ".weight") is not None:
                    op.wgrad(xin, dzc, G(nm + ".weight"), G(nm + ".bias"), h, w)
            if d_bsum is None:
                d_bsum = op.dgrad(dzc, P[op_name + ".weight"], h, w)
            else:
                op.dgrad(dzc, P[op_name + ".weight"], h, w, dx=d_bsum, accumulate=True)
        if d_bsum is None:
            return None
        outs, g_next = S["bott_outs"], None
        for j in reversed(range(self.depth)):
            name = "bottleneck.bottleneck%d.0" % (j + 1)
            dz = K.lrelu_bwd(d_bsum, outs[j], SLOPE, dy2=g_next)
            if G(name + ".weight") is not None:
                self.ops[name].wgrad(S[name], dz, G(name + ".weight"), G(name + ".bias"), h, w)
            g_next = self.ops[name].dgrad(dz, P[name + ".weight"], h, w)
        # every weight gradient except the encoder's has been launched: a data-parallel trainer starts their
        # all-reduce here, under the encoder's backward pass (train_step.py)
        cb = self.after_deep_grads
        if cb is not None:
            K.flush_wgrad_reduces()      # (the hook reads the gradients launched so far)
            cb()
        dA, dB = g_next, None
        for i in reversed(range(nb)):
            hi, wi = H >> i, W >> i
            idx = S["pool%d" % i]
            if i > 0:
                c1 = "encoder.conv1_%d.0" % (i + 1)
                y, res_prev, t = S[c1]
                # the pool's scatter is read in place: no full-resolution d_t (three quarters zeros) is written
                dzc = K.lrelu_bwd_pooled(dA, idx, t, SLOPE, g2=dB)
                if G(c1 + ".weight") is not None:
                    self.ops[c1].wgrad(y, dzc, G(c1 + ".weight"), G(c1 + ".bias"), hi, wi, x2=res_prev)
                d_y = torch.empty((y.t if isinstance(y, TA) else y).shape, dtype=torch.float32, device=dzc.device)
                dB = torch.empty(res_prev.shape, dtype=torch.float32, device=dzc.device)
                self.ops[c1].dgrad(dzc, P[c1 + ".weight"], hi, wi, dx=d_y, dx2=dB)
                pooled = None
            else:
                d_y, pooled, dB = None, (dA, dB, idx), None
            dA, _ = self._dc_bwd(P, G, "encoder.encoder%d" % (i + 1), d_y, d_skips[i], hi, wi, S, i > 0 or need_dx,
                                 pooled=pooled)
        return dA


class _SegFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, *params):
        if not x.is_cuda:
            raise RuntimeError("Segmentation_model_Point runs on HIP devices only (no CPU fallback)")
        x = x.contiguous().float()
        P = module._tensor_dict()
        logits, verts, S = module._engine.forward(P, x, module.training)
        if module.training:
            module._bump_batches_tracked()
        ctx.module, ctx.S, ctx.P = module, S, P
        if getattr(module, "_keep_state", False):     # tests (shared-routing backward checks)
            module._last_S = S
        ctx.set_materialize_grads(False)
        out2 = S.get("out2")
        ctx.layout = (verts is not None, out2 is not None)
        outs = [logits] + ([verts] if verts is not None else []) + ([out2] if out2 is not None else [])
        return outs[0] if len(outs) == 1 else tuple(outs)

    @staticmethod
    def backward(ctx, d_logits, *rest):
        rest = list(rest)
        d_verts = rest.pop(0) if ctx.layout[0] else None
        d_out2 = rest.pop(0) if ctx.layout[1] else None
        with K.deferred_wgrad_reduces():      # (PCUDA_BATCH_REDUCE=1: the layers' split-K reduces in two launches instead of 43)
            dx = ctx.module._engine.backward(ctx.P, ctx.S, d_logits, d_verts, ctx.needs_input_grad[1], d_out2=d_out2)
        ctx.S = None
        return (None, dx) + (None,) * (len(ctx.needs_input_grad) - 2)


# ============================================================================ module tree (reference names)
class Encoder(nn.Module):
    def __init__(self, filters=32, in_channels=3, n_block=4, kernel_size=(3, 3), batch_norm=True, padding='same'):
        super().__init__()
        self.filter = filters
        pad = kernel_size[0] // 2 if padding == 'same' else 0
        for i in range(n_block):
            out_ch = filters * 2 ** i
            in_ch = in_channels if i == 0 else filters * 2 ** (i - 1)
            model = [Conv2d(in_ch, out_ch, kernel_size, padding=pad), LeakyReLU(inplace=True)]
            if batch_norm:
                model += [BatchNorm2d(out_ch)]
            model += [Conv2d(out_ch, out_ch, kernel_size, padding=pad), LeakyReLU(inplace=True)]
            if batch_norm:
                model += [BatchNorm2d(out_ch)]
            self.add_module('encoder%d' % (i + 1), nn.Sequential(*model))
            self.add_module('conv1_%d' % (i + 1), nn.Sequential(Conv2d(in_ch * 3, out_ch, 1), LeakyReLU(inplace=True)))


class Bottleneck(nn.Module):
    def __init__(self, filters=32, n_block=4, depth=4, kernel_size=(3, 3)):
        super().__init__()
        out_ch, in_ch = filters * 2 ** n_block, filters * 2 ** (n_block - 1)
        for i in range(depth):
            d = 2 ** i
            self.add_module('bottleneck%d' % (i + 1), nn.Sequential(
                Conv2d(in_ch, out_ch, kernel_size, padding=d, dilation=d), LeakyReLU(inplace=True)))
            if i == 0:
                in_ch = out_ch


class PointNet(nn.Module):
    def __init__(self, num_points=300, fc_inch=81, conv_inch=512, ext=False):
        super().__init__()
        self.num_points = num_points
        self.ReLU = LeakyReLU(inplace=True)
        if ext:                                                       # unet.py:81-83
            self.conv1 = Conv2d(conv_inch, conv_inch * 2, kernel_size=3, padding=1)
            self.conv2 = Conv2d(conv_inch * 2, conv_inch, kernel_size=3, padding=1)
        self.final_conv = Conv2d(conv_inch, self.num_points, kernel_size=6)
        self.final_fc = Linear(fc_inch, 3)
        self._ext = ext


class Decoder(nn.Module):
    def __init__(self, filters=32, n_block=4, kernel_size=(3, 3), batch_norm=True, padding='same', drop=False):
        super().__init__()
        if drop:
            raise NotImplementedError("Decoder(drop=True) is never used by the reference scripts")
        self.n_block = n_block
        pad = kernel_size[0] // 2 if padding == 'same' else 0
        for i in reversed(range(n_block)):
            out_ch = filters * 2 ** i
            in_ch = 2 * out_ch
            self.add_module('decoder1_%d' % (i + 1), nn.Sequential(
                Marker("UpsamplingNearest2d(scale_factor=2), folded into the next conv"),
                Conv2d(in_ch, out_ch, kernel_size, padding=pad)))
            model = [Conv2d(in_ch, out_ch, kernel_size, padding=pad), LeakyReLU(inplace=True)]
            if batch_norm:
                model += [BatchNorm2d(out_ch)]
            model += [Conv2d(out_ch, out_ch, kernel_size, padding=pad), LeakyReLU(inplace=True)]
            if batch_norm:
                model += [BatchNorm2d(out_ch)]
            self.add_module('decoder2_%d' % (i + 1), nn.Sequential(*model))


class Segmentation_model_Point(nn.Module):
    """unet.py:165-233.  forward(x, features_out=True) -> (logits, None, verts|None) or logits."""

    def __init__(self, filters=32, in_channels=3, n_block=4, bottleneck_depth=4, n_class=4, pointnet=False,
                 fc_inch=81, heinit=False, multicuda=False, extpn=False, batchnorm=True):
        super().__init__()
        if multicuda:
            raise NotImplementedError("multicuda (2-GPU model split, unet.py:180-192) is replaced by data "
                                      "parallelism: see pointcloududa_amd.parallel")
        self._pointnet = pointnet
        self.encoder = Encoder(filters=filters, in_channels=in_channels, n_block=n_block, batch_norm=batchnorm)
        self.bottleneck = Bottleneck(filters=filters, n_block=n_block, depth=bottleneck_depth)
        if pointnet:
            self.pointNet = PointNet(num_points=300, fc_inch=fc_inch, conv_inch=512 * filters // 32, ext=extpn)
        self.decoder = Decoder(filters=filters, n_block=n_block, drop=False, batch_norm=batchnorm)
        self.classifier = Conv2d(filters, n_class, kernel_size=(1, 1))
        self._initialize_weights(heinit=heinit)
        self._multicuda = False
        self._build_engine(filters, in_channels, n_block, bottleneck_depth, n_class, pointnet, fc_inch, extpn, batchnorm)

    def _build_engine(self, filters, in_channels, n_block, bottleneck_depth, n_class, pointnet, fc_inch, extpn, batchnorm,
                      feature_dis=False):
        self._engine = _SegEngine(filters, in_channels, n_block, bottleneck_depth, n_class, pointnet, fc_inch, extpn,
                                  batchnorm=batchnorm, feature_dis=feature_dis)
        for op in self._engine.ops.values():
            op.owner = self

    def _initialize_weights(self, heinit=False):                      # unet.py:194-208
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                std = float(np.sqrt(2 / float(np.prod(m.weight.size()[1:])))) if heinit else 0.02
                m.weight.data.normal_(0.0, std)
                if m.bias is not None:
                    m.bias.data.zero_()

    def tomulticuda(self):
        return None

    # -- plumbing for the fused autograd node
    def _tensor_dict(self):
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        return d

    def _bump_batches_tracked(self):
        flat = getattr(self, "_flat_tracked", None)
        if flat is not None:
            flat.add_(1)
            return
        for k, b in self.named_buffers():
            if k.endswith("num_batches_tracked"):
                b.add_(1)

    def forward(self, x, features_out=True, print_shape=False):
        params = [p for p in self.parameters()]
        out = _SegFn.apply(self, x, *params)
        out = list(out) if isinstance(out, tuple) else [out]
        logits = out.pop(0)
        verts = out.pop(0) if self._pointnet else None
        self._out2 = out.pop(0) if out else None
        if print_shape:
            print("output: {}".format(logits.size()))
            if verts is not None:
                print("pointcloud: {}".format(verts.size()))
        if features_out:
            return logits, None, verts
        return logits


class Segmentation_model(Segmentation_model_Point):
    """unet.py:139-162 (not used by the reference scripts): the same network without the point head.
    ``feature_dis=True`` adds ``classifier2`` -- a 1x1 convolution with 512 input channels, as hard-wired in the reference
    (:147-148; i.e. filters = 32, n_block = 4) -- on the bottleneck output; forward then returns (logits, output2, None)."""

    def __init__(self, filters=32, in_channels=3, n_block=4, bottleneck_depth=4, n_class=4, feature_dis=False):
        super().__init__(filters=filters, in_channels=in_channels, n_block=n_block,
                         bottleneck_depth=bottleneck_depth, n_class=n_class, pointnet=False)
        self._feature_dis = bool(feature_dis)
        for m in self.modules():      # the reference's Segmentation_model has no init pass (unet.py:140-150): torch's defaults
            if isinstance(m, nn.Conv2d):
                m.reset_parameters()
        if feature_dis:
            if filters * 2 ** n_block != 512:
                raise ValueError("feature_dis: classifier2 takes 512 channels (unet.py:148); the bottleneck has %d"
                                 % (filters * 2 ** n_block))
            self.classifier2 = Conv2d(512, n_class, kernel_size=(1, 1))
            # (the reference builds classifier2 AFTER its constructor has no init pass: default nn.Conv2d init stays)
            self._build_engine(filters, in_channels, n_block, bottleneck_depth, n_class, False, 81, False, True,
                               feature_dis=True)

    def forward(self, x, features_out=True):
        logits, _, _ = super().forward(x, features_out=True)
        if features_out:
            return logits, (self._out2 if self._feature_dis else None), None
        return logits
