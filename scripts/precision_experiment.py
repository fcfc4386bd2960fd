"""Item: "2-product" precision modes, measured.  CPU emulation of the MFMA operand roundings on the full-size
segmenter (seg_full256 configuration: filters 32, 256x256, batch 2): every convolution's input and/or weight is
rounded the way a mode's operand split would round it, accumulation stays fp32 (as in the MFMA), and logits /
vertices / loss are compared with the fp32 oracle.  Bar (north star): 1e-3.

modes: x3      = act hi+lo, weight hi+lo, lo*lo dropped  (the parity mode of the HIP kernels)
       a1w2_*  = ONE activation operand, weight hi+lo    (2 MFMA per product instead of 3)
       a2w1_*  = activation hi+lo, ONE weight operand
       a1w1_*  = single operands (throughput mode)
"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nets as ON, losses as OL
from oracle.synth import synth_batch

torch.set_num_threads(8)
_real_conv = F.conv2d


def rnd(t, dt):
    return t.to(dt).to(torch.float32)


def split(t, dt):
    hi = rnd(t, dt)
    return hi, rnd(t - hi, dt)


def make_conv(mode):
    if mode == "fp32":
        return _real_conv
    kind, dts = mode.rsplit("_", 1) if "_" in mode else (mode, "bf16")
    dt = torch.bfloat16 if dts == "bf16" else torch.float16

    def conv(x, w, b=None, **kw):
        xh, xl = split(x, dt)
        wh, wl = split(w, dt)
        if kind == "x3":
            y = _real_conv(xh, wh, None, **kw) + _real_conv(xh, wl, None, **kw) + _real_conv(xl, wh, None, **kw)
        elif kind == "a1w2":
            y = _real_conv(xh, wh + wl, None, **kw)
        elif kind == "a2w1":
            y = _real_conv(xh + xl, wh, None, **kw)
        elif kind == "a1w1":
            y = _real_conv(xh, wh, None, **kw)
        else:
            raise ValueError(mode)
        return y if b is None else y + b.view(1, -1, 1, 1)
    return conv


def run(mode, cfg, params, img, mask, vert):
    F.conv2d = make_conv(mode)
    try:
        with torch.no_grad():
            lo, ve = ON.seg_forward(params, torch.from_numpy(img), cfg, training=True)
            m, j = OL.seg_loss_sigmoid(lo, torch.from_numpy(mask))
            l3 = OL.batch_nn_loss(ve, torch.from_numpy(vert))
    finally:
        F.conv2d = _real_conv
    return lo, ve, float(m + j), float(l3)


def main():
    cfg = ON.SegCfg(filters=32, in_channels=1, n_class=4, pointnet=True, fc_inch=121)
    params = ON.make_params(ON.seg_param_shapes(cfg), 500)
    img, mask, vert, _, _ = synth_batch(2, 1, 4, 256, seed=501)
    ref = run("fp32", cfg, params, img, mask, vert)
    print("fp32: |logits|max %.3f  loss %.6f  nnloss %.6f" % (float(ref[0].abs().max()), ref[2], ref[3]))
    print("%-10s %12s %12s %12s %12s" % ("mode", "max|dlogit|", "max|dvert|", "dloss_rel", "dnn_rel"))
    for mode in sys.argv[1:] or ["x3", "a1w2_bf16", "a1w2_fp16", "a2w1_bf16", "a2w1_fp16", "a1w1_bf16", "a1w1_fp16"]:
        lo, ve, l, l3 = run(mode, cfg, params, img, mask, vert)
        print("%-10s %12.3e %12.3e %12.3e %12.3e" % (mode, float((lo - ref[0]).abs().max()), float((ve - ref[1]).abs().max()),
                                                     abs(l - ref[2]) / abs(ref[2]), abs(l3 - ref[3]) / abs(ref[3])), flush=True)


if __name__ == "__main__":
    main()
