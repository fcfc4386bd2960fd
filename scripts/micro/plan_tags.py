"""plans (tile shape, pixel blocks, kernel family, LDS) and times of single layers: scripts/conv_micro.py's cases through
the library's per-launch event profile.   usage: python3 scripts/micro/plan_tags.py d2 d2f d3 d3e ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloududa_amd import kernels as K
dev = torch.device("cuda", 0)
sys.argv, names = sys.argv[:1] + ["__none__"], sys.argv[1:]
CASES = {}
src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conv_micro.py")).read()
exec(src[src.index("CASES = {"):src.index("which = ")])
for name in names:
    n, cin, cout, h, w, k, s, p, d, up = CASES[name]
    op = K.ConvOp(cin, cout, k, stride=s, pad=p, dil=d, in_up=up)
    x = torch.randn(n, cin, h, w, device=dev); wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    oh, ow = op.out_hw(h, w)
    gz = torch.randn(n, cout, oh, ow, device=dev); dw = torch.zeros_like(wt)
    for _ in range(3):
        op.forward(x, wt, None, 0.2, h, w); op.dgrad(gz, wt, h, w); op.wgrad(x, gz, dw, None, h, w)
    torch.cuda.synchronize()
    K.prof_enable(True); K.prof_reset()
    for _ in range(5):
        op.forward(x, wt, None, 0.2, h, w); op.dgrad(gz, wt, h, w); op.wgrad(x, gz, dw, None, h, w)
    torch.cuda.synchronize()
    path = "/tmp/plan_%s.csv" % name
    K.prof_dump(path); K.prof_enable(False)
    print("==", name)
    print(open(path).read())
