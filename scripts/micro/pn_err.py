import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from conftest import GOLD
from test_networks_gpu import _load, rel_err, _strided
from oracle import nets as ON
from pointcloududa_amd.networks import PointNetCls
from pointcloududa_amd.utils import loss as L
dev = torch.device("cuda", 0)
for tag, ft, ext in [("pncls", False, False), ("pncls_ft_ext", True, True)]:
    g = np.load(os.path.join(GOLD, tag + ".npz"))
    seed, b = int(g["seed"]), int(g["b"])
    params = ON.make_params(ON.pointnet_cls_param_shapes(ft, ext=ext), seed)
    model = _load(PointNetCls(feature_transform=ft, ext=ext, drop=0.0), params, dev)
    rng = np.random.default_rng(seed + 1)
    x = torch.from_numpy(rng.random((b, 3, 300), dtype=np.float32)).to(dev).requires_grad_(True)
    y, trans, tf = model(x)
    loss = L.bce_logits_const(y, 0.0); loss.backward()
    print(tag, "b", b, "y", rel_err(y, g["y"]), "trans", rel_err(trans, g["trans"]), "loss", abs(float(loss) - float(g["loss"])), "dx", rel_err(x.grad, g["dx"]),
          "tf", rel_err(_strided(tf), g["trans_feat_s"]) if ft else None, [k for k in g.files if "spread" in k][:3])
