"""validation / inference forward throughput: eval-mode segmenter forward + label map + Dice,
B=32, 256x256, random-init weights, synthetic batch"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from pointcloududa_amd import validate as V
dev = torch.device("cuda", 0)
wl = B.WORKLOADS["full_uda"]
tr = B.build_trainer(wl, dev, seed=0)
img_a, mask_a, vert_a, img_b, vert_b = B.synth_device_batch(32, 256, 4, seed=100, dev=dev)
gen = tr.gen.eval()
for _ in range(3):
    r = V.valid_batch(gen, img_a, mask_a, vert_a)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 20
for _ in range(n):
    r = V.valid_batch(gen, img_a, mask_a, vert_a)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("validation batch (eval forward + losses + labels + dice): %.2f ms/batch of 32 -> %.0f img/s; dice %.4f loss %.4f" % (
    dt * 1e3, 32 / dt, float(r["dice"]), float(r["loss"])))
