"""CPU experiment (oracle only): how well-conditioned are the reference network's gradients?
Adds white noise of a given relative size to conv outputs of the fp32 oracle and reports the change
of logits / gradients against an fp64 run.  Result: a threshold at ~1e-6 -- one
max-pool argmax / LeakyReLU sign flip at the 8x8 level moves some weight gradients by 1-2 % -- so
gradient parity between ANY two fp32-class implementations is ~1e-2, while outputs stay at 1e-4."""
import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, torch.nn.functional as F
from oracle import nets as ON, losses as OL
from oracle.synth import synth_batch
torch.manual_seed(0)
cfg = ON.SegCfg(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
params = ON.make_params(ON.seg_param_shapes(cfg), 100)
img, mask, vert, _, _ = synth_batch(2, 1, 4, 128, seed=101)
def run(noise, dtype=torch.float32):
    p = {k: (v.to(dtype).clone().requires_grad_(True) if ON.is_trainable(k) else (v.to(dtype).clone() if v.is_floating_point() else v.clone())) for k, v in params.items()}
    x = torch.from_numpy(img).to(dtype).requires_grad_(True)
    orig = F.conv2d
    def noisy(*a, **k):
        y = orig(*a, **k)
        if noise > 0:
            y = y + noise * y.detach().abs().max() * torch.randn_like(y) * 0.3
        return y
    ON.F.conv2d = noisy
    try:
        lo, ve = ON.seg_forward(p, x, cfg, True)
    finally:
        ON.F.conv2d = orig
    m, j = OL.seg_loss_sigmoid(lo, torch.from_numpy(mask))
    (m + j + OL.batch_nn_loss(ve, torch.from_numpy(vert).to(dtype))).backward()
    return lo.detach(), x.grad, {k: v.grad for k, v in p.items() if ON.is_trainable(k) and v.grad is not None}
def rel(a, b): return float((a.double()-b.double()).abs().max()/b.double().abs().max())
lo0, dx0, g0 = run(0, torch.float64)
for nz in (0.0, 1e-6, 1e-5):
    lo, dx, g = run(nz)
    worst = max(rel(g[k], g0[k]) for k in g0)
    wk = max(g0, key=lambda k: rel(g[k], g0[k]))
    print("noise %g: logits err %.2e  dx err %.2e  worst param grad err %.2e (%s)  enc1.5.w %.2e" % (nz, rel(lo, lo0), rel(dx, dx0), worst, wk, rel(g["encoder.encoder1.5.weight"], g0["encoder.encoder1.5.weight"])))
print("---- without the point head / NN loss")
cfg = ON.SegCfg(filters=4, in_channels=1, n_class=4, pointnet=False)
params = ON.make_params(ON.seg_param_shapes(cfg), 100)
def run2(noise, dtype=torch.float32, only=None, seedn=0):
    torch.manual_seed(seedn)
    p = {k: (v.to(dtype).clone().requires_grad_(True) if ON.is_trainable(k) else (v.to(dtype).clone() if v.is_floating_point() else v.clone())) for k, v in params.items()}
    x = torch.from_numpy(img).to(dtype).requires_grad_(True)
    orig = F.conv2d; cnt = [0]
    def noisy(*a, **k):
        y = orig(*a, **k); cnt[0] += 1
        if noise > 0 and (only is None or cnt[0] == only):
            y = y + noise * y.detach().abs().max() * torch.randn_like(y) * 0.3
        return y
    ON.F.conv2d = noisy
    try:
        lo, ve = ON.seg_forward(p, x, cfg, True)
    finally:
        ON.F.conv2d = orig
    m, j = OL.seg_loss_sigmoid(lo, torch.from_numpy(mask))
    (m + j).backward()
    return lo.detach(), x.grad, {k: v.grad for k, v in p.items() if ON.is_trainable(k) and v.grad is not None}
lo0, dx0, g0 = run2(0, torch.float64)
for nz, only in ((0.0, None), (1e-6, None), (1e-6, 1), (1e-6, 25), (1e-6, 12)):
    lo, dx, g = run2(nz, only=only)
    worst = max(rel(g[k], g0[k]) for k in g0); wk = max(g0, key=lambda k: rel(g[k], g0[k]))
    print("noise %g only=%s: logits %.2e dx %.2e worst grad %.2e (%s)" % (nz, only, rel(lo, lo0), rel(dx, dx0), worst, wk))
# norms of gradient vs typical term: is dx cancellation-dominated?
print("dx max", float(dx0.abs().max()), "mean abs", float(dx0.abs().mean()))
print("---- scaling of the effect of layer-1 noise")
for nz in (1e-8, 1e-7, 1e-6, 1e-5, 1e-4):
    lo, dx, g = run2(nz, only=1)
    print("noise %g: logits %.2e dx %.2e enc4.3.w %.2e dec2_1.3.w %.2e" % (nz, rel(lo, lo0), rel(dx, dx0), rel(g["encoder.encoder4.3.weight"], g0["encoder.encoder4.3.weight"]), rel(g["decoder.decoder2_1.3.weight"], g0["decoder.decoder2_1.3.weight"])))
