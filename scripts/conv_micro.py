"""time single conv layers (forward, dgrad, wgrad) of the benchmark's shapes; PCUDA_DBG bits switch
parts of igemm_kernel off for timing experiments"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloududa_amd import kernels as K
dev = torch.device("cuda", 0)
K.set_precision(os.environ.get("PREC", "bf16x3"))
CASES = {  # name: (n, cin, cout, h, w, k, s, p, d, up)
    "g32": (32, 32, 32, 256, 256, 3, 1, 1, 1, False),
    "g64": (32, 64, 64, 128, 128, 3, 1, 1, 1, False),
    "g6432": (32, 64, 32, 256, 256, 3, 1, 1, 1, False), "g12864": (32, 128, 64, 128, 128, 3, 1, 1, 1, False),
    "g3264": (32, 32, 64, 128, 128, 3, 1, 1, 1, False),
    "g128": (32, 128, 128, 64, 64, 3, 1, 1, 1, False),
    "g256": (32, 256, 256, 32, 32, 3, 1, 1, 1, False),
    "b512": (32, 512, 512, 16, 16, 3, 1, 1, 1, False),
    "b512d4": (32, 512, 512, 16, 16, 3, 1, 4, 4, False), "b512d8": (32, 512, 512, 16, 16, 3, 1, 8, 8, False),
    "d1": (32, 4, 64, 256, 256, 4, 2, 2, 1, False), "md1": (32, 4, 64, 224, 224, 4, 2, 2, 1, False),
    "d2": (32, 64, 128, 129, 129, 4, 2, 2, 1, False),
    "d3": (32, 128, 256, 65, 65, 4, 2, 2, 1, False),
    # even-sized twins of the discriminator layers ("Round 3" item 10: what they measure is tile fit)
    "d2e": (32, 64, 128, 128, 128, 4, 2, 2, 1, False), "d2f": (32, 64, 128, 126, 126, 4, 2, 2, 1, False),
    "d3e": (32, 128, 256, 64, 64, 4, 2, 2, 1, False), "d3f": (32, 128, 256, 62, 62, 4, 2, 2, 1, False),
    "d4e": (32, 256, 512, 32, 32, 4, 2, 2, 1, False), "d4f": (32, 256, 512, 30, 30, 4, 2, 2, 1, False),
    "d4": (32, 256, 512, 33, 33, 4, 2, 2, 1, False),
    # the reference's real MS-CMRSeg shape (224x224: train_mscmrseg.py:412): segmenter levels 224 / 112 / 56 / 28 / 14,
    # discriminator maps 113 / 57 / 29
    "m32": (32, 32, 32, 224, 224, 3, 1, 1, 1, False), "m64": (32, 64, 64, 112, 112, 3, 1, 1, 1, False),
    "m128": (32, 128, 128, 56, 56, 3, 1, 1, 1, False), "m256": (32, 256, 256, 28, 28, 3, 1, 1, 1, False),
    "mb512": (32, 512, 512, 14, 14, 3, 1, 1, 1, False), "mb512d8": (32, 512, 512, 14, 14, 3, 1, 8, 8, False),
    "md2": (32, 64, 128, 113, 113, 4, 2, 2, 1, False), "md3": (32, 128, 256, 57, 57, 4, 2, 2, 1, False),
    "md4": (32, 256, 512, 29, 29, 4, 2, 2, 1, False),
    # the discriminators' stride-2 4x4 layers written as 2x2 / stride-1 layers over a space-to-depth input (4 cin channels at
    # half resolution): what the same MACs cost in that form on today's kernels (s2d* = d2 / d3 / d4)
    "s2d2": (32, 256, 128, 67, 67, 2, 1, 0, 1, False), "s2d3": (32, 512, 256, 35, 35, 2, 1, 0, 1, False),
    "s2d4": (32, 1024, 512, 19, 19, 2, 1, 0, 1, False),
    # the residual 1x1 convolutions behind the concatenation (unet.py:41-47) and the classifier: p* (p = pointwise)
    "p96": (32, 96, 64, 128, 128, 1, 1, 0, 1, False), "p192": (32, 192, 128, 64, 64, 1, 1, 0, 1, False),
    "p384": (32, 384, 256, 32, 32, 1, 1, 0, 1, False), "pcls": (32, 32, 4, 256, 256, 1, 1, 0, 1, False),
    "mp96": (32, 96, 64, 112, 112, 1, 1, 0, 1, False), "mp192": (32, 192, 128, 56, 56, 1, 1, 0, 1, False),
    "mp384": (32, 384, 256, 28, 28, 1, 1, 0, 1, False),
    # the decoder's up-convolutions (unet.py:85: Upsample(x2) -> 3x3; h, w = the upsampled size, the input is stored at half of it)
    "u6432": (32, 64, 32, 256, 256, 3, 1, 1, 1, True), "u12864": (32, 128, 64, 128, 128, 3, 1, 1, 1, True),
    "u256128": (32, 256, 128, 64, 64, 3, 1, 1, 1, True), "u512256": (32, 512, 256, 32, 32, 3, 1, 1, 1, True),
}
which = sys.argv[1:] or list(CASES)
for name in which:
    n, cin, cout, h, w, k, s, p, d, up = CASES[name]
    n = int(os.environ.get("MICRO_N", n))      # (batch override: what a joint source + target launch would cost)
    # the variants the NETWORKS launch (and the default build holds, csrc/variants.h): the segmenter's 3x3 layers have a bias,
    # LeakyReLU(0.01) and BatchNorm partial sums -- the bottleneck (b*/mb*) no BatchNorm; the discriminators' stride-2 layers
    # (d*/md*) no bias, LeakyReLU(0.2), no statistics
    disc, bott = name.lstrip("m").startswith("d") or name.startswith("s2d"), name.lstrip("m").startswith("b")
    stats, slope = not (disc or bott or up or name.lstrip("m").startswith("p")), (0.2 if disc else 0.01)
    if os.environ.get("MICRO_NOSTATS") == "1": stats = False
    op = K.ConvOp(cin, cout, k, stride=s, pad=p, dil=d, in_up=up)
    # MICRO_PAD=<floats>: plane pitch of every activation tensor padded by that many floats (are power-of-two plane strides -- all
    # channel planes of a pixel on one memory channel -- what the mixed read / write streams of the full-resolution level pay for?)
    PAD = int(os.environ.get("MICRO_PAD", "0"))
    def padded(nn, cc, hh, ww, fill="randn"):
        big = (torch.randn if fill == "randn" else torch.zeros)(nn, cc, hh * ww + PAD, device=dev)
        return big[:, :, :hh * ww].view(nn, cc, hh, ww) if PAD else big.view(nn, cc, hh, ww)
    x = padded(n, cin, h // 2 if up else h, w // 2 if up else w); wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    b = None if disc else torch.zeros(cout, device=dev)
    oh, ow = op.out_hw(h, w)
    gz = padded(n, cout, oh, ow)
    yout, dxout = padded(n, cout, oh, ow, 'zeros'), padded(n, cin, h // 2 if up else h, w // 2 if up else w, 'zeros')
    # MICRO_DATA=zeros: all-zero operands, zerow: zero weights only -- the SAME instruction streams on operands that do not
    # toggle the multipliers: what the kernels' clock is under (round 6: profiles/r06_experiment_power_cap.txt)
    if os.environ.get("MICRO_DATA") == "zeros": x.zero_(); wt.zero_(); gz.zero_()
    if os.environ.get("MICRO_DATA") == "zerow": wt.zero_()
    dw = torch.zeros_like(wt)
    fl = 2.0 * n * oh * ow * cout * cin * k * k
    def t(fn, reps=int(os.environ.get("MICRO_REPS", 10))):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    tf = t(lambda: op.forward(x, wt, b, slope, h, w, want_stats=stats, out=yout if PAD else None))
    td = t(lambda: op.dgrad(gz, wt, h, w, dx=dxout if (PAD and not up) else None))
    tw = t(lambda: op.wgrad(x, gz, dw, b, h, w))
    print("%-5s fwd %7.3f ms %6.1f TF | dgrad %7.3f ms %6.1f TF | wgrad %7.3f ms %6.1f TF" % (name, tf*1e3, fl/tf/1e12, td*1e3, fl/td/1e12, tw*1e3, fl/tw/1e12), flush=True)
