import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import nets as ON
from pointcloududa_amd.networks import Segmentation_model_Point, UncertaintyDiscriminator
from pointcloududa_amd.utils import loss as L
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
p = {k: v.clone() for k, v in params.items()}
# CPU reference, stepwise
def dc(blk, xin):
    z0 = F.leaky_relu(F.conv2d(xin, p[blk+".0.weight"], p[blk+".0.bias"], padding=1), 0.01)
    y0 = F.batch_norm(z0, None, None, p[blk+".2.weight"], p[blk+".2.bias"], True)
    z1 = F.leaky_relu(F.conv2d(y0, p[blk+".3.weight"], p[blk+".3.bias"], padding=1), 0.01)
    y1 = F.batch_norm(z1, None, None, p[blk+".5.weight"], p[blk+".5.bias"], True)
    return z0, y0, z1, y1
cur = x; res = None; skips = []
for i in range(4):
    blk = "encoder.encoder%d" % (i+1)
    z0, y0, z1, y1 = dc(blk, cur)
    xs, x2s, a0, st0, a1, st1 = S[blk]
    print(blk, "a0", rel(a0, z0), "mean0", rel(st0.mean, z0.mean((0,2,3))), "invstd0", rel(st0.invstd, 1/torch.sqrt(z0.var((0,2,3), unbiased=False)+1e-5)),
          "a1", rel(a1, z1), "y1", rel(a1*st1.scale[None,:,None,None]+st1.shift[None,:,None,None], y1))
    skips.append(y1)
    out = y1
    if i > 0:
        c1 = "encoder.conv1_%d.0" % (i+1)
        out = F.leaky_relu(F.conv2d(torch.cat([out, res], 1), p[c1+".weight"], p[c1+".bias"]), 0.01)
        print(c1, rel(S[c1][2], out))
    out = F.max_pool2d(out, 2); res = out; cur = out
o = cur; tot = None
for j in range(4):
    name = "bottleneck.bottleneck%d.0" % (j+1); d = 2**j
    o = F.leaky_relu(F.conv2d(o, p[name+".weight"], p[name+".bias"], padding=d, dilation=d), 0.01)
    print(name, rel(S["bott_outs"][j], o))
    tot = o if tot is None else tot + o
print("bsum", rel(S["head"][0], tot))
hc = F.leaky_relu(F.conv2d(tot, p["pointNet.final_conv.weight"], p["pointNet.final_conv.bias"]), 0.01)
print("head conv", rel(S["head"][1], hc))
v = F.linear(hc.reshape(2, 300, -1), p["pointNet.final_fc.weight"], p["pointNet.final_fc.bias"])
print("verts", rel(verts, v))
out = tot
for i in reversed(range(4)):
    up = "decoder.decoder1_%d.1" % (i+1)
    u = F.conv2d(F.interpolate(out, scale_factor=2, mode="nearest"), p[up+".weight"], p[up+".bias"], padding=1)
    blk = "decoder.decoder2_%d" % (i+1)
    xs, x2s, a0, st0, a1, st1 = S[blk]
    print(up, rel(x2s, u))
    z0, y0, z1, y1 = dc(blk, torch.cat([skips[i], u], 1))
    print(blk, "a0", rel(a0, z0), "a1", rel(a1, z1))
    out = y1
lg = F.conv2d(out, p["classifier.weight"], p["classifier.bias"])
print("logits", rel(logits, lg))
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests/golden/seg_small.npz"))
print("cpu-ref vs golden", rel(lg, torch.from_numpy(g["logits"])), "hip vs golden", rel(logits, torch.from_numpy(g["logits"])))
m2 = Segmentation_model_Point(**cfg_kw); m2.load_state_dict({k: v.clone() for k, v in params.items()}); m2 = m2.to(dev).train()
xx = x.to(dev).requires_grad_(True)
lo2, _, ve2 = m2(xx)
print("module fwd vs golden (before bwd)", rel(lo2, torch.from_numpy(g["logits"])))
wl = torch.from_numpy(rng.normal(0, 1, (2, 4, 128, 128)).astype(np.float32)).to(dev)
wv = torch.from_numpy(rng.normal(0, 1, (2, 300, 3)).astype(np.float32)).to(dev)
torch.autograd.backward([lo2, ve2], [wl / lo2.numel(), wv / ve2.numel()])
torch.cuda.synchronize()
print("module fwd vs golden (after bwd)", rel(lo2, torch.from_numpy(g["logits"])))
print("dx", rel(xx.grad, torch.from_numpy(g["dx"])))
for k, pp in m2.named_parameters():
    if "g/" + k in g:
        print(k, rel(pp.grad, torch.from_numpy(g["g/" + k])))
