"""per-phase cycle sums of igemm_pipe_kernel / igemm8_kernel (PCUDA_DBG=128) for single layers, launched the way the networks
launch them (the discriminators' layers: no bias, LeakyReLU 0.2, no BatchNorm partial sums -- the variants the default build holds).
Needs a library with the stamps compiled in: make -C pointcloududa_amd/csrc clean all CLK=1 (or OUT=<path> and PCUDA_LIB=<path>)."""
import os, sys, ctypes
os.environ["PCUDA_DBG"] = "128"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointcloududa_amd import kernels as K
from pointcloududa_amd import _lib
lib = _lib.lib()
dev = torch.device("cuda", 0)
K.set_precision(os.environ.get("PREC", "bf16x3"))
CASES = {"g32": (32, 32, 32, 256, 256, 3, 1, 1, 1), "g64": (32, 64, 64, 128, 128, 3, 1, 1, 1),
         "g128": (32, 128, 128, 64, 64, 3, 1, 1, 1), "g256": (32, 256, 256, 32, 32, 3, 1, 1, 1),
         "d2": (32, 64, 128, 129, 129, 4, 2, 2, 1), "d4": (32, 256, 512, 33, 33, 4, 2, 2, 1)}
names = ["barrier(top)", "commit X", "W issue+X issue", "W commit/copy", "barrier(W)", "MFMA taps", "epilogue", "loop tail"]
buf = (ctypes.c_ulonglong * 8)()
for name in sys.argv[1:] or list(CASES):
    n, cin, cout, h, w, k, s, p, d = CASES[name]
    op = K.ConvOp(cin, cout, k, stride=s, pad=p, dil=d)
    x = torch.randn(n, cin, h, w, device=dev); wt = torch.randn(cout, cin, k, k, device=dev) * 0.05
    disc = name.startswith("d")
    b = None if disc else torch.zeros(cout, device=dev)
    for which in ("fwd", "dgrad"):
        oh, ow = op.out_hw(h, w)
        gz = torch.randn(n, cout, oh, ow, device=dev)
        fn = (lambda: op.forward(x, wt, b, 0.2 if disc else 0.01, h, w, want_stats=not disc)) if which == "fwd" else (lambda: op.dgrad(gz, wt, h, w))
        fn(); torch.cuda.synchronize(); lib.pcuda_debug_read_clocks(buf)
        fn(); torch.cuda.synchronize(); lib.pcuda_debug_read_clocks(buf)
        tot = sum(buf)
        print(name, which, "total wave0 cycles (sum over WGs) %.3g" % tot, "|", K.last_kernel())
        for i in [7, 0, 1, 2, 3, 4, 5, 6]:
            print("   %-18s %5.1f%%" % (names[(i + 1) % 8] if False else ["barrier(top)", "commit X", "W issue+X issue", "W commit/copy", "barrier(W)", "MFMA taps", "epilogue", "loop tail"][i], 100.0 * buf[i] / max(tot, 1)))
