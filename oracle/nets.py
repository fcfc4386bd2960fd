"""Functional CPU restatement of the three reference networks (oracle; test-only).

Every function takes a flat ``dict`` of tensors whose keys are exactly the
reference modules' ``state_dict`` keys, so the same dictionary can be
``load_state_dict``-ed into the reference (``oracle/make_golden.py`` does that)
or into the HIP-backed modules of ``pointcloududa_amd.networks``.

Reference being restated (file:line are relative to /root/reference/src):
  networks/unet.py:7-51     Encoder           -> _encoder
  networks/unet.py:54-73    Bottleneck        -> _bottleneck
  networks/unet.py:76-96    PointNet (head)   -> _point_head
  networks/unet.py:100-136  Decoder           -> _decoder
  networks/unet.py:165-233  Segmentation_model_Point -> seg_forward
  networks/GAN.py:89-144    UncertaintyDiscriminator -> disc_forward
  networks/PointNetCls.py:11-214  STN3d/STNkd/PointNetfeat/PointNetCls -> pointnet_cls_forward
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

# --------------------------------------------------------------------------- #
# shared-routing anchors (tests only)
# --------------------------------------------------------------------------- #
# The reference networks' gradients are discontinuous in their activations (LeakyReLU / ReLU sign, max-pool and
# max-over-points argmax): two fp32-class implementations whose pre-activations differ by 1e-6 take different
# branches on a few elements and then disagree on some weight gradients by percents.  To compare BACKWARD passes
# at 1e-4 a test installs an anchor: a callable ``anchor(tag, z) -> z'`` invoked on every routing-relevant
# intermediate (conv / linear outputs, normalisation outputs); it returns ``z + (z_other - z).detach()``, i.e. the
# VALUES the implementation under test produced, while autograd still differentiates this restatement.  With no
# anchor installed (the default, and always in make_golden.py) the functions below are the plain restatement.
_ANCHOR = None


class anchored:
    """``with anchored(fn): ...`` installs ``fn(tag, tensor) -> tensor`` for the forward passes inside the block"""

    def __init__(self, fn):
        self.fn = fn

    def __enter__(self):
        global _ANCHOR
        self.prev, _ANCHOR = _ANCHOR, self.fn
        return self

    def __exit__(self, *exc):
        global _ANCHOR
        _ANCHOR = self.prev
        return False


def _anc(tag: str, z):
    return z if _ANCHOR is None else _ANCHOR(tag, z)


# --------------------------------------------------------------------------- #
# configuration + parameter inventories
# --------------------------------------------------------------------------- #
@dataclass(frozen=True)
class SegCfg:
    """ctor arguments of Segmentation_model_Point (unet.py:169-170)."""
    filters: int = 32
    in_channels: int = 3
    n_block: int = 4
    bottleneck_depth: int = 4
    n_class: int = 4
    pointnet: bool = False
    fc_inch: int = 81
    extpn: bool = False
    batchnorm: bool = True
    feature_dis: bool = False      # Segmentation_model(feature_dis=True): classifier2 on the bottleneck output (unet.py:147-158)


def seg_param_shapes(cfg: SegCfg) -> Dict[str, Tuple[int, ...]]:
    """state_dict inventory of Segmentation_model_Point, in registration order."""
    f, shapes = cfg.filters, {}

    def conv(name, co, ci, k):
        shapes[name + ".weight"] = (co, ci, k, k)
        shapes[name + ".bias"] = (co,)

    def bn(name, c):
        shapes[name + ".weight"] = (c,)
        shapes[name + ".bias"] = (c,)
        shapes[name + ".running_mean"] = (c,)
        shapes[name + ".running_var"] = (c,)
        shapes[name + ".num_batches_tracked"] = ()

    for i in range(cfg.n_block):                       # unet.py:12-33
        co = f * 2 ** i
        ci = cfg.in_channels if i == 0 else f * 2 ** (i - 1)
        blk = "encoder.encoder%d" % (i + 1)
        conv(blk + ".0", co, ci, 3)
        if cfg.batchnorm:
            bn(blk + ".2", co)
            conv(blk + ".3", co, co, 3)
            bn(blk + ".5", co)
        else:
            conv(blk + ".2", co, co, 3)
        conv("encoder.conv1_%d.0" % (i + 1), co, ci * 3, 1)
    co = f * 2 ** cfg.n_block                          # unet.py:57-65
    ci = f * 2 ** (cfg.n_block - 1)
    for j in range(cfg.bottleneck_depth):
        conv("bottleneck.bottleneck%d.0" % (j + 1), co, ci, 3)
        ci = co
    if cfg.pointnet:                                   # unet.py:77-87
        cin = 512 * f // 32
        if cfg.extpn:
            conv("pointNet.conv1", cin * 2, cin, 3)
            conv("pointNet.conv2", cin, cin * 2, 3)
        conv("pointNet.final_conv", 300, cin, 6)
        shapes["pointNet.final_fc.weight"] = (3, cfg.fc_inch)
        shapes["pointNet.final_fc.bias"] = (3,)
    for i in reversed(range(cfg.n_block)):             # unet.py:108-126
        co = f * 2 ** i
        conv("decoder.decoder1_%d.1" % (i + 1), co, 2 * co, 3)
        blk = "decoder.decoder2_%d" % (i + 1)
        conv(blk + ".0", co, 2 * co, 3)
        if cfg.batchnorm:
            bn(blk + ".2", co)
            conv(blk + ".3", co, co, 3)
            bn(blk + ".5", co)
        else:
            conv(blk + ".2", co, co, 3)
    conv("classifier", cfg.n_class, f, 1)
    if cfg.feature_dis:
        conv("classifier2", cfg.n_class, 512, 1)
    return shapes


def disc_param_shapes(in_channel: int = 2, ext: bool = False) -> Dict[str, Tuple[int, ...]]:
    """state_dict inventory of UncertaintyDiscriminator (GAN.py:95-107): no biases."""
    s = {"conv1.weight": (64, in_channel, 4, 4), "conv2.weight": (128, 64, 4, 4),
         "conv3.weight": (256, 128, 4, 4), "conv4.weight": (512, 256, 4, 4)}
    if ext:
        s["conv4_2.weight"] = (1024, 512, 3, 3)
        s["conv4_3.weight"] = (256, 1024, 3, 3)
        s["conv5.weight"] = (1, 256, 4, 4)
    else:
        s["conv5.weight"] = (1, 512, 4, 4)
    return s


def _bn1d_shapes(s, name, c, affine=True):
    if affine:
        s[name + ".weight"] = (c,)
        s[name + ".bias"] = (c,)
    s[name + ".running_mean"] = (c,)
    s[name + ".running_var"] = (c,)
    s[name + ".num_batches_tracked"] = ()


def _stn_shapes(s, prefix, cin, kout, with_in):
    """STN3d (PointNetCls.py:16-36) / STNkd (:67-83) inventories."""
    for n, co, ci in (("conv1", 64, cin), ("conv2", 128, 64), ("conv3", 1024, 128)):
        s[prefix + n + ".weight"] = (co, ci, 1)
        s[prefix + n + ".bias"] = (co,)
    for n, co, ci in (("fc1", 512, 1024), ("fc2", 256, 512), ("fc3", kout, 256)):
        s[prefix + n + ".weight"] = (co, ci)
        s[prefix + n + ".bias"] = (co,)
    for n, c in (("bn1", 64), ("bn2", 128), ("bn3", 1024), ("bn4", 512), ("bn5", 256)):
        _bn1d_shapes(s, prefix + n, c)
    if with_in:
        for n, c in (("in1", 64), ("in2", 128), ("in3", 1024), ("in4", 512), ("in5", 256)):
            _bn1d_shapes(s, prefix + n, c, affine=False)


def pointnet_cls_param_shapes(feature_transform=False, in_channel=3, dim=3, ext=False,
                              kernel_size=1) -> Dict[str, Tuple[int, ...]]:
    """state_dict inventory of PointNetCls (PointNetCls.py:170-183, 104-133)."""
    s: Dict[str, Tuple[int, ...]] = {}
    _stn_shapes(s, "feat.stn.", dim, 9, with_in=True)
    k = kernel_size
    if ext:
        chain = (("conv1", 8, in_channel, "bn1"), ("conv1_1", 64, 8, "bn1_1"),
                 ("conv2", 128, 64, "bn2"), ("conv2_1", 256, 128, "bn2_1"),
                 ("conv3", 512, 256, "bn3"), ("conv3_1", 1024, 512, "bn3_1"))
        for n, co, ci, b in chain:
            s["feat.%s.weight" % n] = (co, ci, k)
            s["feat.%s.bias" % n] = (co,)
            _bn1d_shapes(s, "feat." + b, co)
    else:
        for n, co, ci in (("conv1", 64, in_channel), ("conv2", 128, 64), ("conv3", 1024, 128)):
            s["feat.%s.weight" % n] = (co, ci, k)
            s["feat.%s.bias" % n] = (co,)
        for n, c in (("bn1", 64), ("bn2", 128), ("bn3", 1024)):
            _bn1d_shapes(s, "feat." + n, c)
    if feature_transform:
        _stn_shapes(s, "feat.fstn.", 64, 64 * 64, with_in=False)
    for n, co, ci in (("fc1", 512, 1024), ("fc2", 256, 512), ("fc3", 1, 256)):
        s[n + ".weight"] = (co, ci)
        s[n + ".bias"] = (co,)
    _bn1d_shapes(s, "bn1", 512)
    _bn1d_shapes(s, "bn2", 256)
    _bn1d_shapes(s, "in1", 512, affine=False)
    _bn1d_shapes(s, "in2", 256, affine=False)
    return s


def make_params(shapes: Dict[str, Tuple[int, ...]], seed: int, *, std: float = 0.05,
                bias_std: float = 0.02) -> Params:
    """Portable deterministic parameters from ``numpy.random.default_rng(seed)``.

    Not the reference's initialiser (that one draws from torch's RNG,
    unet.py:194-208); fixtures use this stream so the very same weights can be
    regenerated on the GPU box.  Norm scales are drawn around 1, running_var
    stays positive, ``num_batches_tracked`` starts at 0.
    """
    rng = np.random.default_rng(seed)
    out: Params = {}
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            out[k] = torch.from_numpy(rng.uniform(0.5, 1.5, shp).astype(np.float32))
        elif k.endswith("running_mean"):
            out[k] = torch.from_numpy(rng.normal(0, 0.1, shp).astype(np.float32))
        elif len(shp) == 1 and k.endswith(".weight"):          # norm gamma
            out[k] = torch.from_numpy(rng.normal(1.0, 0.1, shp).astype(np.float32))
        elif k.endswith(".bias"):
            out[k] = torch.from_numpy(rng.normal(0, bias_std, shp).astype(np.float32))
        else:
            fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else 1
            s = min(std, math.sqrt(2.0 / max(fan_in, 1)))
            out[k] = torch.from_numpy(rng.normal(0, s, shp).astype(np.float32))
    return out


def is_trainable(key: str) -> bool:
    return not (key.endswith("running_mean") or key.endswith("running_var")
                or key.endswith("num_batches_tracked"))


# --------------------------------------------------------------------------- #
# segmenter
# --------------------------------------------------------------------------- #
def _bn2d(p: Params, name: str, x, training: bool):
    """torch BatchNorm2d defaults: momentum 0.1, eps 1e-5 (unet.py:26)."""
    y = F.batch_norm(x, p[name + ".running_mean"], p[name + ".running_var"],
                     p[name + ".weight"], p[name + ".bias"], training, 0.1, 1e-5)
    if training:
        p[name + ".num_batches_tracked"] += 1
    return y


def _conv(p: Params, name: str, x, **kw):
    return _anc(name, F.conv2d(x, p[name + ".weight"], p.get(name + ".bias"), **kw))


def _double_conv(p: Params, blk: str, x, cfg: SegCfg, training: bool):
    """conv3x3 -> LeakyReLU(0.01) -> BN -> conv3x3 -> LeakyReLU -> BN (unet.py:23-30)."""
    x = F.leaky_relu(_conv(p, blk + ".0", x, padding=1), 0.01)
    if cfg.batchnorm:
        x = _bn2d(p, blk + ".2", x, training)
        x = F.leaky_relu(_conv(p, blk + ".3", x, padding=1), 0.01)
        x = _bn2d(p, blk + ".5", x, training)
    else:
        x = F.leaky_relu(_conv(p, blk + ".2", x, padding=1), 0.01)
    return x


def _encoder(p: Params, x, cfg: SegCfg, training: bool):
    """unet.py:35-51.  Block 1 skips its 1x1 conv (``if i > 1``), so
    ``encoder.conv1_1`` never runs and never receives a gradient."""
    skips: List[torch.Tensor] = []
    out, res = x, None
    for i in range(cfg.n_block):
        out = _double_conv(p, "encoder.encoder%d" % (i + 1), out, cfg, training)
        skips.append(out)
        if i > 0:
            out = torch.cat([out, res], 1)
            out = F.leaky_relu(_conv(p, "encoder.conv1_%d.0" % (i + 1), out), 0.01)
        out = F.max_pool2d(out, 2)
        res = out
    return out, skips


def _bottleneck(p: Params, x, cfg: SegCfg):
    """unet.py:67-73: dilations 1,2,4,8..., outputs chained AND summed."""
    total, out = None, x
    for j in range(cfg.bottleneck_depth):
        d = 2 ** j
        out = F.leaky_relu(_conv(p, "bottleneck.bottleneck%d.0" % (j + 1), out,
                                 padding=d, dilation=d), 0.01)
        total = out if total is None else total + out
    return total


def _point_head(p: Params, x, cfg: SegCfg):
    """unet.py:89-96: 6x6 valid conv -> LeakyReLU -> flatten -> Linear(fc_inch, 3)."""
    if cfg.extpn:
        x = F.leaky_relu(_conv(p, "pointNet.conv1", x, padding=1), 0.01)
        x = F.leaky_relu(_conv(p, "pointNet.conv2", x, padding=1), 0.01)
    x = F.leaky_relu(_conv(p, "pointNet.final_conv", x), 0.01)
    x = x.reshape(x.shape[0], x.shape[1], -1)
    return _anc("pointNet.final_fc", F.linear(x, p["pointNet.final_fc.weight"], p["pointNet.final_fc.bias"]))


def _decoder(p: Params, x, skips: List[torch.Tensor], cfg: SegCfg, training: bool):
    """unet.py:128-136: nearest x2 -> conv3x3 (no act) -> cat(skip, .) -> double conv."""
    out = x
    for i in reversed(range(cfg.n_block)):
        out = F.interpolate(out, scale_factor=2, mode="nearest")
        out = _conv(p, "decoder.decoder1_%d.1" % (i + 1), out, padding=1)
        out = torch.cat([skips[i], out], 1)
        out = _double_conv(p, "decoder.decoder2_%d" % (i + 1), out, cfg, training)
    return out


def seg_forward(p: Params, x, cfg: SegCfg, training: bool = True):
    """Segmentation_model_Point.forward (unet.py:210-233) -> (logits, verts|None).

    BatchNorm running statistics in ``p`` are updated in place when training.
    """
    out, skips = _encoder(p, x, cfg, training)
    bott = _bottleneck(p, out, cfg)
    verts = _point_head(p, bott, cfg) if cfg.pointnet else None
    out = _decoder(p, bott, skips, cfg, training)
    logits = _conv(p, "classifier", out)
    if cfg.feature_dis:      # Segmentation_model.forward (unet.py:152-162) -> (logits, output2)
        return logits, _conv(p, "classifier2", bott)
    return logits, verts


# --------------------------------------------------------------------------- #
# image discriminator
# --------------------------------------------------------------------------- #
def disc_forward(p: Params, x, ext: bool = False):
    """UncertaintyDiscriminator.forward (GAN.py:131-144): 4x4 s2 p2 convs, LeakyReLU(0.2)."""
    for n in ("conv1", "conv2", "conv3", "conv4"):
        x = F.leaky_relu(_anc(n, F.conv2d(x, p[n + ".weight"], None, stride=2, padding=2)), 0.2)
    if ext:
        x = F.leaky_relu(_anc("conv4_2", F.conv2d(x, p["conv4_2.weight"], None, stride=2, padding=1)), 0.2)
        x = F.leaky_relu(_anc("conv4_3", F.conv2d(x, p["conv4_3.weight"], None, stride=2, padding=1)), 0.2)
    return _anc("conv5", F.conv2d(x, p["conv5.weight"], None, stride=2, padding=2))


def output_disc_forward(p: Params, x, softmax: bool = False):
    """OutputDiscriminator.forward (GAN.py:77-86): bilinear resize to 224x224 (align_corners=True), optional channel
    softmax, then the same five 4x4 s2 p2 layers."""
    x = F.interpolate(x, size=(224, 224), mode="bilinear", align_corners=True)    # nn.UpsamplingBilinear2d
    if softmax:
        x = F.softmax(x, dim=1)
    return disc_forward(p, x, ext=False)


def fc_disc_param_shapes() -> Dict[str, Tuple[int, ...]]:
    """Discriminator (GAN.py:7-18)"""
    s, cin = {}, 24576
    for i, co in enumerate((4096, 2048, 1024, 1)):
        s["fc%d.weight" % (i + 1)] = (co, cin)
        s["fc%d.bias" % (i + 1)] = (co,)
        cin = co
    return s


def fc_disc_forward(p: Params, x):
    """Discriminator.forward (GAN.py:42-49)"""
    for i in (1, 2, 3):
        x = F.leaky_relu(_anc("fc%d" % i, F.linear(x, p["fc%d.weight" % i], p["fc%d.bias" % i])), 0.2)
    return _anc("fc4", F.linear(x, p["fc4.weight"], p["fc4.bias"]))


# --------------------------------------------------------------------------- #
# point-cloud discriminator
# --------------------------------------------------------------------------- #
def _norm1d(p: Params, name_bn: str, name_in: Optional[str], x, training: bool, batched: bool):
    """BatchNorm1d when batch > 1, InstanceNorm1d(track_running_stats) when batch == 1
    (PointNetCls.py:40-55, 207-212).  x is [B,C] or [B,C,L]."""
    if batched:
        y = F.batch_norm(x, p[name_bn + ".running_mean"], p[name_bn + ".running_var"],
                         p[name_bn + ".weight"], p[name_bn + ".bias"], training, 0.1, 1e-5)
        if training:
            p[name_bn + ".num_batches_tracked"] += 1
        return _anc(name_bn, y)
    # InstanceNorm1d on a 2-D [1,C] input treats it as unbatched [C=1? no: (C,L)=(1,C)]:
    # torch raises for mismatched features unless the tensor is 3-D.  The reference only
    # reaches this path with batch == 1; restated literally.
    y = F.instance_norm(x, p[name_in + ".running_mean"], p[name_in + ".running_var"],
                        None, None, training, 0.1, 1e-5)
    if training:
        p[name_in + ".num_batches_tracked"] += 1
    return _anc(name_in, y)


def _stn(p: Params, pre: str, x, k: int, training: bool, has_in: bool):
    """STN3d (PointNetCls.py:38-63) / STNkd (:85-102): returns [B,k,k] = fc3(.) + I."""
    b = x.shape[0]
    batched = (b > 1) or not has_in
    h = x
    for i in (1, 2, 3):
        h = _anc(pre + "conv%d" % i, F.conv1d(h, p[pre + "conv%d.weight" % i], p[pre + "conv%d.bias" % i]))
        h = F.relu(_norm1d(p, pre + "bn%d" % i, pre + "in%d" % i, h, training, batched))
    h = h.max(dim=2)[0].reshape(-1, 1024)
    for i, j in ((1, 4), (2, 5)):
        h = _anc(pre + "fc%d" % i, F.linear(h, p[pre + "fc%d.weight" % i], p[pre + "fc%d.bias" % i]))
        h = F.relu(_norm1d(p, pre + "bn%d" % j, pre + "in%d" % j, h, training, batched))
    h = F.linear(h, p[pre + "fc3.weight"], p[pre + "fc3.bias"])
    h = _anc(pre + "fc3", h + torch.eye(k, dtype=h.dtype).reshape(1, k * k))
    return h.reshape(-1, k, k)


def pointnet_cls_forward(p: Params, x, *, feature_transform=False, sample_transform=True,
                         ext=False, drop: float = 0.3, training: bool = True,
                         drop_mask: Optional[torch.Tensor] = None):
    """PointNetCls.forward (PointNetCls.py:204-214) on x[B,3,N] -> (logit[B,1], trans, trans_feat).

    ``drop_mask`` ([B,256], values 0 or 1/(1-p)) replaces the reference's
    ``nn.Dropout(p)`` RNG draw so that parity runs are deterministic; ``None`` with
    ``drop == 0`` (or eval) means identity.
    """
    b = x.shape[0]
    trans = trans_feat = None
    if sample_transform:                                       # :138-142
        trans = _stn(p, "feat.stn.", x, 3, training, has_in=True)
        x = torch.bmm(x.transpose(2, 1), trans).transpose(2, 1)

    def cbr(conv, bn, h, relu=True):
        h = _anc("feat." + conv, F.conv1d(h, p["feat.%s.weight" % conv], p["feat.%s.bias" % conv]))
        h = _norm1d(p, "feat." + bn, None, h, training, True)
        return F.relu(h) if relu else h

    h = cbr("conv1", "bn1", x)
    if ext:
        h = cbr("conv1_1", "bn1_1", h)
    if feature_transform:                                      # :147-151
        trans_feat = _stn(p, "feat.fstn.", h, 64, training, has_in=False)
        h = torch.bmm(h.transpose(2, 1), trans_feat).transpose(2, 1)
    h = cbr("conv2", "bn2", h)
    if ext:
        h = cbr("conv2_1", "bn2_1", h)
    h = cbr("conv3", "bn3", h, relu=False)                     # :159 (no ReLU)
    if ext:
        h = cbr("conv3_1", "bn3_1", h)
    h = h.max(dim=2)[0].reshape(-1, 1024)                      # :162-163

    batched = b > 1
    h = _anc("fc1", F.linear(h, p["fc1.weight"], p["fc1.bias"]))
    h = F.relu(_norm1d(p, "bn1", "in1", h, training, batched))
    h = _anc("fc2", F.linear(h, p["fc2.weight"], p["fc2.bias"]))
    if training and drop > 0:
        if drop_mask is None:
            raise ValueError("oracle needs an explicit drop_mask when drop > 0 in training")
        h = h * drop_mask
    h = F.relu(_norm1d(p, "bn2", "in2", h, training, batched))
    h = _anc("fc3", F.linear(h, p["fc3.weight"], p["fc3.bias"]))
    return h, trans, trans_feat


def feature_transform_regularizer(trans):
    """PointNetCls.py:217-224: mean Frobenius norm of (T T^T - I)."""
    d = trans.shape[1]
    eye = torch.eye(d, dtype=trans.dtype)[None]
    return torch.mean(torch.norm(torch.bmm(trans, trans.transpose(2, 1)) - eye, dim=(1, 2)))
