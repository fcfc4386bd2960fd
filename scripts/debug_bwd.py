import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import nets as ON
from pointcloududa_amd import kernels as K
from pointcloududa_amd.kernels import TA
from pointcloududa_amd.networks import Segmentation_model_Point
dev = torch.device("cuda", 0)
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(1e-30, b.abs().max()))
cfg_kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
cfg = ON.SegCfg(**cfg_kw)
params = ON.make_params(ON.seg_param_shapes(cfg), 100)
m = Segmentation_model_Point(**cfg_kw); m.load_state_dict({k: v.clone() for k, v in params.items()}); m = m.to(dev).train()
rng = np.random.default_rng(101)
x = torch.from_numpy(rng.random((2, 1, 128, 128), dtype=np.float32))
P = m._tensor_dict()
logits, verts, S = m._engine.forward(P, x.to(dev), True)
blk = "decoder.decoder2_1"
xs, x2s, a0, st0, a1, st1 = S[blk]
p = {k: v.clone() for k, v in params.items()}
# CPU: block input = cat[skip (normalised), u]
skip = (xs.t * xs.scale[None, :, None, None] + xs.shift[None, :, None, None]).cpu()
xin = torch.cat([skip, x2s.cpu()], 1).requires_grad_(True)
w0 = p[blk+".0.weight"].requires_grad_(True); w3 = p[blk+".3.weight"].requires_grad_(True)
z0p = F.conv2d(xin, w0, p[blk+".0.bias"], padding=1); z0p.retain_grad()
z0 = F.leaky_relu(z0p, 0.01)
y0 = F.batch_norm(z0, None, None, p[blk+".2.weight"], p[blk+".2.bias"], True); y0.retain_grad()
z1p = F.conv2d(y0, w3, p[blk+".3.bias"], padding=1); z1p.retain_grad()
z1 = F.leaky_relu(z1p, 0.01)
y1 = F.batch_norm(z1, None, None, p[blk+".5.weight"], p[blk+".5.bias"], True)
gy = torch.from_numpy(rng.normal(0, 1, y1.shape).astype(np.float32))
y1.backward(gy)
print("a0", rel(a0, z0), "a1", rel(a1, z1))
dg = torch.zeros(4, device=dev); db = torch.zeros(4, device=dev)
dz1 = K.bn_backward(gy.to(dev), a1, st1, P[blk+".5.weight"], dg, db, act_slope=0.01, accumulate=False)
print("dz1", rel(dz1, z1p.grad))
op3 = m._engine.ops[blk+".3"]; op0 = m._engine.ops[blk+".0"]
dw3 = torch.zeros(4, 4, 3, 3, device=dev); db3 = torch.zeros(4, device=dev)
op3.wgrad(TA(a0, st0.scale, st0.shift), dz1, dw3, db3, 128, 128, accumulate=False)
print("dw3", rel(dw3, w3.grad), "dw3 with ref dz", end=" ")
op3.wgrad(TA(a0, st0.scale, st0.shift), z1p.grad.to(dev), dw3, db3, 128, 128, accumulate=False)
print(rel(dw3, w3.grad))
d_y0 = op3.dgrad(dz1, P[blk+".3.weight"], 128, 128)
print("d_y0", rel(d_y0, y0.grad), "with ref dz", rel(op3.dgrad(z1p.grad.to(dev), P[blk+".3.weight"], 128, 128), y0.grad))
dz0 = K.bn_backward(d_y0, a0, st0, P[blk+".2.weight"], dg, db, act_slope=0.01, accumulate=False)
print("dz0", rel(dz0, z0p.grad))
dw0 = torch.zeros(4, 8, 3, 3, device=dev); db0 = torch.zeros(4, device=dev)
op0.wgrad(xs, dz0, dw0, db0, 128, 128, x2=x2s, accumulate=False)
print("dw0", rel(dw0, w0.grad))
print("---- full engine backward vs oracle autograd")
pp = {k: (v.clone().requires_grad_(True) if ON.is_trainable(k) else v.clone()) for k, v in params.items()}
lo, ve = ON.seg_forward(pp, x, cfg, training=True)
wl = torch.from_numpy(rng.normal(0, 1, lo.shape).astype(np.float32))
(lo * wl).sum().backward()
for q in m.parameters():
    q.grad = None
m._engine.backward(P, S, wl.to(dev), None, False)
for k in ["classifier.weight", "decoder.decoder2_1.5.weight", "decoder.decoder2_1.3.weight", "decoder.decoder2_1.2.weight",
          "decoder.decoder2_1.0.weight", "decoder.decoder1_1.1.weight", "decoder.decoder2_2.5.weight", "decoder.decoder2_2.3.weight",
          "encoder.encoder1.3.weight", "encoder.encoder1.0.weight"]:
    print(k, rel(P[k].grad, pp[k].grad))
print("---- classifier dgrad check")
opc = m._engine.ops["classifier"]
dl = wl.to(dev)
d_cur = opc.dgrad(dl, P["classifier.weight"], 128, 128)
ref = F.conv_transpose2d(wl, params["classifier.weight"])
print("d_cur", rel(d_cur, ref))
# element-wise relative profile
e = (d_cur.cpu() - ref).abs(); print("max abs err", float(e.max()), "ref max", float(ref.abs().max()), "mean abs err", float(e.mean()), "mean abs ref", float(ref.abs().mean()))
# now scale d_logits down like the test (1/numel)
d_small = opc.dgrad(dl / dl.numel(), P["classifier.weight"], 128, 128)
print("d_cur small", rel(d_small, ref / dl.numel()))
dz1s = K.bn_backward(d_small, a1, st1, P[blk+".5.weight"], dg, db, act_slope=0.01, accumulate=False)
y1b = F.batch_norm(F.leaky_relu(z1p.detach().requires_grad_(True), 0.01), None, None, p[blk+".5.weight"], p[blk+".5.bias"], True)
print("---- CPU block (HIP forward inputs) with gy = d_cur vs oracle full-network grad")
for t in (xin, w0, w3):
    t.grad = None
z0p = F.conv2d(xin, w0, p[blk+".0.bias"], padding=1)
z0 = F.leaky_relu(z0p, 0.01)
y0 = F.batch_norm(z0, None, None, p[blk+".2.weight"], p[blk+".2.bias"], True)
z1p = F.conv2d(y0, w3, p[blk+".3.bias"], padding=1)
z1 = F.leaky_relu(z1p, 0.01)
y1 = F.batch_norm(z1, None, None, p[blk+".5.weight"], p[blk+".5.bias"], True)
y1.backward(d_cur.cpu())
print("w3 block-vs-oracle", rel(w3.grad, pp[blk+".3.weight"].grad), " hip-vs-block", rel(P[blk+".3.weight"].grad, w3.grad))
# sensitivity: oracle forward in float64
pd = {k: (v.double().clone().requires_grad_(True) if ON.is_trainable(k) else v.double().clone()) for k, v in params.items()}
lo64, _ = ON.seg_forward(pd, x.double(), cfg, training=True)
(lo64 * wl.double()).sum().backward()
print("oracle f32 vs f64 grad", rel(pp[blk+".3.weight"].grad, pd[blk+".3.weight"].grad), " hip vs f64", rel(P[blk+".3.weight"].grad, pd[blk+".3.weight"].grad))
print("enc1.0: oracle f32 vs f64", rel(pp["encoder.encoder1.0.weight"].grad, pd["encoder.encoder1.0.weight"].grad), " hip vs f64", rel(P["encoder.encoder1.0.weight"].grad, pd["encoder.encoder1.0.weight"].grad))
print("---- which input perturbs the block gradient?")
def dc(blk_, xin_):
    z0_ = F.leaky_relu(F.conv2d(xin_, p[blk_+".0.weight"], p[blk_+".0.bias"], padding=1), 0.01)
    y0_ = F.batch_norm(z0_, None, None, p[blk_+".2.weight"], p[blk_+".2.bias"], True)
    z1_ = F.leaky_relu(F.conv2d(y0_, p[blk_+".3.weight"], p[blk_+".3.bias"], padding=1), 0.01)
    return F.batch_norm(z1_, None, None, p[blk_+".5.weight"], p[blk_+".5.bias"], True)
with torch.no_grad():
    cur = x; res = None; skips = []
    for i in range(4):
        y1_ = dc("encoder.encoder%d" % (i+1), cur); skips.append(y1_); out = y1_
        if i > 0:
            c1 = "encoder.conv1_%d.0" % (i+1)
            out = F.leaky_relu(F.conv2d(torch.cat([out, res], 1), p[c1+".weight"], p[c1+".bias"]), 0.01)
        out = F.max_pool2d(out, 2); res = out; cur = out
    o = cur; tot = None
    for j in range(4):
        name = "bottleneck.bottleneck%d.0" % (j+1); d = 2**j
        o = F.leaky_relu(F.conv2d(o, p[name+".weight"], p[name+".bias"], padding=d, dilation=d), 0.01)
        tot = o if tot is None else tot + o
    out = tot
    for i in reversed(range(4)):
        up = "decoder.decoder1_%d.1" % (i+1)
        u_cpu = F.conv2d(F.interpolate(out, scale_factor=2, mode="nearest"), p[up+".weight"], p[up+".bias"], padding=1)
        if i > 0:
            out = dc("decoder.decoder2_%d" % (i+1), torch.cat([skips[i], u_cpu], 1))
skip_cpu = skips[0]
print("skip hip vs cpu", rel(skip, skip_cpu), " u hip vs cpu", rel(x2s, u_cpu))
def block_grad(sk, uu, gy_):
    w3_ = p[blk+".3.weight"].detach().clone().requires_grad_(True)
    xin_ = torch.cat([sk, uu], 1)
    z0_ = F.leaky_relu(F.conv2d(xin_, p[blk+".0.weight"].detach(), p[blk+".0.bias"], padding=1), 0.01)
    y0_ = F.batch_norm(z0_, None, None, p[blk+".2.weight"], p[blk+".2.bias"], True)
    z1_ = F.leaky_relu(F.conv2d(y0_, w3_, p[blk+".3.bias"], padding=1), 0.01)
    y1_ = F.batch_norm(z1_, None, None, p[blk+".5.weight"], p[blk+".5.bias"], True)
    y1_.backward(gy_)
    return w3_.grad
gref = pp[blk+".3.weight"].grad
gy_c = d_cur.cpu()
print("cpu,cpu", rel(block_grad(skip_cpu, u_cpu, gy_c), gref))
print("hip skip", rel(block_grad(skip.detach(), u_cpu, gy_c), gref))
print("hip u", rel(block_grad(skip_cpu, x2s.cpu(), gy_c), gref))
print("cpu + noise 1e-4", rel(block_grad(skip_cpu, u_cpu + 1e-4 * u_cpu.abs().max() * torch.randn_like(u_cpu), gy_c), gref))
