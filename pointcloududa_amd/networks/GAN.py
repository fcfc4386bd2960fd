"""HIP-backed drop-in for the reference's ``src/networks/GAN.py``.

``UncertaintyDiscriminator`` (GAN.py:89-144) is the class both image discriminators of the train
scripts instantiate (d1 on logits / probabilities, d2 on the entropy map).  Its five 4x4 stride-2
convolutions run on the MFMA implicit-GEMM kernels; the adversarial gradient that flows back to
the segmenter is the stride-2 *transposed* convolution (``conv2d_dgrad``: one launch per output ROW
parity, the two column parities paired in one row tile).  The other classes of the file keep their signatures and
parameter names.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from .. import kernels as K
from ..kernels import ConvOp
from ._holders import Conv2d, LeakyReLU, Linear, Marker, ensure_grad


def _chain_forward(module, x, weights, biases, outs=None):
    """the layers in turn; ``outs``: optional preallocated output tensor per layer.  -> (acts, sizes, out)"""
    nl = len(module._chain)
    acts, sizes = [x], []
    h, w = x.shape[2], x.shape[3]
    cur = x
    for li, (name, op) in enumerate(module._chain):
        slope = module._slope if li < nl - 1 else 1.0
        sizes.append((h, w))
        wt = weights[li].view(op.cout, op.cin, op.k, op.k)
        dst = None if outs is None else outs[li]
        if li == 0 and module._fold1 is not None and not (cur.is_contiguous() and K.d1_forward_direct(op, cur.shape[0], h, w)):
            # a handful of input channels: unfold the taps into channels and run the layer as a 1x1 convolution
            # whose 32-deep reduction chunks are full (16 taps x 4 of 32 channels otherwise)
            unfolded = K.unfold_taps(cur, op.k, op.stride, op.pad, op.dil)
            oh, ow = op.out_hw(h, w)
            cur, _, _ = module._fold1.forward(unfolded, wt.view(wt.shape[0], -1, 1, 1), biases[0], slope, oh, ow, out=dst)
        else:
            cur, _, _ = op.forward(cur, wt, biases[li], slope, h, w, out=dst)
        h, w = op.out_hw(h, w)
        acts.append(cur)
    return acts, sizes, cur


def _joint_buffers(module, x, room):
    """per layer one output tensor for (1 + room) batches of x's size: a cached pass writes the LAST batch slot, the
    batches filled in later (``forward_fill``) the ones in front, and ``replay`` walks all of them as one batch"""
    b, h, w = x.shape[0], x.shape[2], x.shape[3]
    full = []
    for name, op in module._chain:
        h, w = op.out_hw(h, w)
        full.append(torch.empty(((1 + room) * b, op.cout, h, w), dtype=torch.float32, device=x.device))
    return full


class _ConvChainFn(torch.autograd.Function):
    """x -> [conv (+bias) -> LeakyReLU(slope)]* -> conv (+bias)   over a list of Conv2d / Linear holders.
    ``wb``: the chain's weights followed by its biases (None where a layer has none)."""

    @staticmethod
    def forward(ctx, module, x, *wb):
        if not x.is_cuda:
            raise RuntimeError("discriminators run on HIP devices only (no CPU fallback)")
        x = x.contiguous().float()
        nl = len(module._chain)
        weights, biases = wb[:nl], wb[nl:]
        caching, room = getattr(module, "_cache_next", False), getattr(module, "_cache_room", 0)
        full = _joint_buffers(module, x, room) if (caching and room) else None
        b = x.shape[0]
        acts, sizes, cur = _chain_forward(module, x, weights, biases,
                                          None if full is None else [f[room * b:] for f in full])
        ctx.module, ctx.acts, ctx.sizes, ctx.weights, ctx.biases = module, acts, sizes, weights, biases
        if getattr(module, "_keep_acts", False):      # tests (shared-routing backward checks)
            module._last_acts = list(acts)
        if caching:                                   # forward_cached(): this pass can be replayed (replay())
            module._cache_next = False
            module._cache = dict(acts=list(acts), sizes=list(sizes), out=cur, full=full, b=b,
                                 x=[None] * room + [x])
        ctx.set_materialize_grads(False)
        return cur

    @staticmethod
    def backward(ctx, d_out):
        return _chain_backward(ctx, d_out)


class _ConvChainReplayFn(torch.autograd.Function):
    """The output of an EARLIER forward pass of the chain (``forward_cached``), with a backward pass of its own through
    the activations that pass kept: weights unchanged in between, same input -> the forward would recompute the same
    bits.  The train step's discriminator update sees the target batch this way: the frozen adversarial pass of phase 2
    already ran the network on it (train_mscmrseg.py:222-241 and :283-322 call D on the same tensor values).  With
    room for more batches in the cached pass's buffers (``forward_fill``) the node spans all of them as ONE batch."""

    @staticmethod
    def forward(ctx, module, cache, *wb):
        nl = len(module._chain)
        if cache["full"] is not None:
            acts, out = [list(cache["x"])] + cache["full"], cache["full"][-1]
        else:
            acts, out = cache["acts"], cache["out"]
        ctx.module, ctx.acts, ctx.sizes, ctx.weights, ctx.biases = module, acts, cache["sizes"], wb[:nl], wb[nl:]
        ctx.set_materialize_grads(False)
        return out.detach()

    @staticmethod
    def backward(ctx, d_out):
        return _chain_backward(ctx, d_out)


def _chain_backward(ctx, d_out):
    nin = len(ctx.needs_input_grad)
    if d_out is None:
        return (None,) * nin
    module, acts, sizes, weights, biases = ctx.module, ctx.acts, ctx.sizes, ctx.weights, ctx.biases
    nl = len(module._chain)
    dz = d_out.contiguous()
    dx = None
    with K.deferred_wgrad_reduces():      # (PCUDA_BATCH_REDUCE=1: the five layers' split-K reduces in one launch)
        for li in reversed(range(nl)):
            name, op = module._chain[li]
            h, w = sizes[li]
            wt = weights[li].view(op.cout, op.cin, op.k, op.k)
            if ctx.needs_input_grad[2 + li]:
                # (the first layer's weight gradient stays on the k x k kernel: the one-tap form of the weight-gradient
                # kernel stages a tile per tap and measured 0.40 ms against 0.25 ms for this layer)
                db = ensure_grad(biases[li]) if (biases[li] is not None and ctx.needs_input_grad[2 + nl + li]) else None
                dw = ensure_grad(weights[li]).view(op.cout, op.cin, op.k, op.k)
                if isinstance(acts[li], list):      # (joint replay: the network's inputs stay separate tensors)
                    bq = acts[li][0].shape[0]
                    for q, xq in enumerate(acts[li]):
                        op.wgrad(xq, dz[q * bq:(q + 1) * bq], dw, db, h, w)
                else:
                    op.wgrad(acts[li], dz, dw, db, h, w)
            if li > 0:
                # (the LeakyReLU backward of the layer in front rides in the data gradient's epilogue)
                dz = op.dgrad_lrelu(dz, wt, h, w, acts[li], module._slope)
            elif ctx.needs_input_grad[1]:
                dx = op.dgrad(dz, wt, h, w)
    ctx.acts = None
    return (None, dx) + (None,) * (nin - 2)


class _ConvChain(nn.Module):
    _slope = 0.2

    def _build_chain(self, names):
        chain = []
        for n in names:
            m = getattr(self, n)
            if isinstance(m, nn.Linear):        # a fully connected layer = a 1x1 convolution over a 1x1 image
                op = ConvOp(m.in_features, m.out_features, 1)
            else:
                op = ConvOp(m.in_channels, m.out_channels, m.kernel_size[0], stride=m.stride[0], pad=m.padding[0],
                            dil=m.dilation[0])
            op.owner = self
            chain.append((n, op))
        self._chain = chain
        # first layer over <= 8 input channels (4 / 5 class maps, 1- or 3-channel boundary maps): tap-unfolded 1x1 form
        m0 = getattr(self, names[0])
        self._fold1 = None
        if (isinstance(m0, nn.Conv2d) and m0.in_channels <= 8 and m0.kernel_size[0] > 1
                and os.environ.get("PCUDA_NOFOLD", "0") != "1"):
            self._fold1 = ConvOp(m0.in_channels * m0.kernel_size[0] ** 2, m0.out_channels, 1)
            self._fold1.owner = self
            self._fold1.pack_dgrad_with_fwd = False      # the input gradient stays with the k x k operator

    def _init_conv(self, heinit=False):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                std = float(np.sqrt(2 / float(np.prod(m.weight.size()[1:])))) if heinit else 0.02
                m.weight.data.normal_(0.0, std)
                if m.bias is not None:
                    m.bias.data.zero_()

    def forward(self, x):
        ws = [getattr(self, n).weight for n, _ in self._chain]
        bs = [getattr(self, n).bias for n, _ in self._chain]
        return _ConvChainFn.apply(self, x, *(ws + bs))

    def forward_cached(self, x, room=0):
        """``forward`` that keeps its activations for ONE later ``replay()`` (dropped by ``drop_cache()``).
        ``room``: batches of the same size that ``forward_fill`` will put IN FRONT of this one before the replay."""
        self._cache_next, self._cache_room = True, int(room)
        try:
            return self(x)
        finally:
            self._cache_next, self._cache_room = False, 0

    def forward_fill(self, x, slot=0):
        """Run the network on another batch (no autograd node) into slot ``slot`` of the cached pass's buffers."""
        cache = getattr(self, "_cache", None)
        if cache is None or cache["full"] is None or slot >= len(cache["x"]) - 1 or x.shape[0] != cache["b"]:
            raise RuntimeError("forward_fill() needs a cached pass with room for a batch of this size")
        ws = [getattr(self, n).weight for n, _ in self._chain]
        bs = [getattr(self, n).bias for n, _ in self._chain]
        b = cache["b"]
        x = x.detach().contiguous().float()
        with torch.no_grad():
            _chain_forward(self, x, ws, bs, [f[slot * b:(slot + 1) * b] for f in cache["full"]])
        cache["x"][slot] = x

    def replay(self):
        """The cached pass's output -- with room: the outputs of all batches in its buffers, filled ones first -- as a
        new autograd node whose backward pass produces THIS call's parameter gradients (the parameters must not have
        changed since ``forward_cached``)."""
        cache = getattr(self, "_cache", None)
        if cache is None:
            raise RuntimeError("replay() without a cached forward pass")
        if any(t is None for t in cache["x"]):
            raise RuntimeError("replay(): a batch slot of the cached pass was never filled")
        ws = [getattr(self, n).weight for n, _ in self._chain]
        bs = [getattr(self, n).bias for n, _ in self._chain]
        return _ConvChainReplayFn.apply(self, cache, *(ws + bs))

    def drop_cache(self):
        self._cache = None
        self._cache_next = False

    @property
    def can_replay(self):
        return type(self).forward is _ConvChain.forward


class UncertaintyDiscriminator(_ConvChain):
    def __init__(self, in_channel=2, heinit=False, ext=False):
        super().__init__()
        f = [64, 128, 256, 512, 1]
        self.conv1 = Conv2d(in_channel, f[0], kernel_size=4, stride=2, padding=2, bias=False)
        self.conv2 = Conv2d(f[0], f[1], kernel_size=4, stride=2, padding=2, bias=False)
        self.conv3 = Conv2d(f[1], f[2], kernel_size=4, stride=2, padding=2, bias=False)
        self.conv4 = Conv2d(f[2], f[3], kernel_size=4, stride=2, padding=2, bias=False)
        names = ["conv1", "conv2", "conv3", "conv4"]
        if ext:
            self.conv4_2 = Conv2d(f[3], 1024, kernel_size=3, stride=2, padding=1, bias=False)
            self.conv4_3 = Conv2d(1024, f[2], kernel_size=3, stride=2, padding=1, bias=False)
            self.conv5 = Conv2d(f[2], f[4], kernel_size=4, stride=2, padding=2, bias=False)
            names += ["conv4_2", "conv4_3"]
        else:
            self.conv5 = Conv2d(f[3], f[4], kernel_size=4, stride=2, padding=2, bias=False)
        names.append("conv5")
        self.leakyrelu = LeakyReLU(negative_slope=0.2)
        self._ext = ext
        self._init_conv(heinit=heinit)
        self._build_chain(names)


class BoundaryDiscriminator(_ConvChain):
    """GAN.py:147-177: the same chain on a 1-channel input."""
    _in = 1

    def __init__(self):
        super().__init__()
        f = [64, 128, 256, 512, 1]
        cin = self._in
        for i, co in enumerate(f):
            setattr(self, "conv%d" % (i + 1), Conv2d(cin, co, kernel_size=4, stride=2, padding=2, bias=False))
            cin = co
        self.leakyrelu = LeakyReLU(negative_slope=0.2)
        self._init_conv()
        self._build_chain(["conv1", "conv2", "conv3", "conv4", "conv5"])


class BoundaryEntDiscriminator(BoundaryDiscriminator):
    """GAN.py:179-209: 3-channel input."""
    _in = 3


class _BilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        if not x.is_cuda:
            raise RuntimeError("discriminators run on HIP devices only (no CPU fallback)")
        x = x.float()
        ctx.hw = (x.shape[2], x.shape[3])
        return K.bilinear_fwd(x, oh, ow)

    @staticmethod
    def backward(ctx, dy):
        return K.bilinear_bwd(dy, *ctx.hw), None, None


class OutputDiscriminator(_ConvChain):
    """GAN.py:52-86 (not instantiated by the reference's train scripts, which use UncertaintyDiscriminator for d1):
    bilinear resize to 224x224 (align_corners=True), optional channel softmax, then the five 4x4 stride-2 layers."""

    def __init__(self, in_channel=2, softmax=False, init=False):
        super().__init__()
        self._softmax = softmax
        f = [64, 128, 256, 512, 1]
        self.upsample = Marker("UpsamplingBilinear2d(size=(224, 224)): pcuda_bilinear_fwd / _bwd")
        cin = in_channel
        for i, co in enumerate(f):
            setattr(self, "conv%d" % (i + 1), Conv2d(cin, co, kernel_size=4, stride=2, padding=2, bias=False))
            cin = co
        self.leakyrelu = LeakyReLU(negative_slope=0.2)
        if init:
            self._init_conv()
        self._build_chain(["conv1", "conv2", "conv3", "conv4", "conv5"])

    def forward(self, x):
        from ..utils.loss import entropy_map
        x = _BilinearFn.apply(x, 224, 224)
        if self._softmax:
            x = entropy_map(x, "softmax", False, want_prob=True)[1]
        return super().forward(x)


class Discriminator(_ConvChain):
    """GAN.py:7-49: fully connected 24576-4096-2048-1024-1 with LeakyReLU(0.2) (unused by the scripts).  Each layer
    runs as a 1x1 convolution over a 1x1 image on the MFMA kernels (bias and LeakyReLU in the epilogue)."""

    def __init__(self):
        super().__init__()
        f = [4096, 2048, 1024, 1]
        self.fc1 = Linear(24576, f[0])
        self.leakyrelu = LeakyReLU(negative_slope=0.2)
        self.fc2 = Linear(f[0], f[1])
        self.fc3 = Linear(f[1], f[2])
        self.fc4 = Linear(f[2], f[3])
        for m in self.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(0.0, 0.02)
                m.bias.data.zero_()
        self._build_chain(["fc1", "fc2", "fc3", "fc4"])

    def forward(self, x):
        lead = x.shape[:-1]
        y = super().forward(x.reshape(-1, x.shape[-1], 1, 1))
        return y.reshape(*lead, 1)
