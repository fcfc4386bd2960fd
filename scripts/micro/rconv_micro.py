"""go / no-go of the record-form convolution (round-3 review item 2A): the 32->32 layer at 256x256 and the 64->64 layer at
128x128 (n = 32) on csrc/conv_rec.hip (input already in MFMA-record form, LDS-DMA staging, record epilogue + BatchNorm
partial sums) against the shipped NCHW kernel (igemm_pipe_kernel, forward with statistics).   usage: python3 scripts/micro/rconv_micro.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pointcloududa_amd import kernels as K
dev = torch.device("cuda", 0)
CASES = {"g32": (32, 32, 32, 256, 256), "g64": (32, 64, 64, 128, 128), "c64_32": (32, 64, 32, 256, 256),
         "g32_224": (32, 32, 32, 224, 224), "g128": (32, 128, 128, 64, 64), "g256": (32, 256, 256, 32, 32)}
def t(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
# stride-2 4x4 layers of the discriminators (GAN.py:97-105) through the production kernel: NCHW source against record source
DCASES = {"d2": (32, 64, 128, 129), "d3": (32, 128, 256, 65), "d4": (32, 256, 512, 33)}
for name in [a for a in sys.argv[1:] if a in DCASES]:
    n, cin, cout, hw = DCASES[name]
    op = K.ConvOp(cin, cout, 4, stride=2, pad=2)
    x = torch.randn(n, cin, hw, hw, device=dev); wt = torch.randn(cout, cin, 4, 4, device=dev) * 0.05
    oh, ow = op.out_hw(hw, hw)
    gz = torch.randn(n, cout, oh, ow, device=dev)
    xr, gzr = K.rec_from_nchw(x), K.rec_from_nchw(gz)
    fl = 2.0 * n * oh * ow * cout * cin * 16
    tf0 = t(lambda: op.forward(x, wt, None, 0.2, hw, hw)); tf1 = t(lambda: op.forward(K.Rec(xr), wt, None, 0.2, hw, hw))
    td0 = t(lambda: op.dgrad(gz, wt, hw, hw)); td1 = t(lambda: op.dgrad(K.Rec(gzr), wt, hw, hw))
    y0, _, _ = op.forward(x, wt, None, 0.2, hw, hw); y1, _, _ = op.forward(K.Rec(xr), wt, None, 0.2, hw, hw)
    e = float((y0 - y1).abs().max() / y0.abs().max())
    print("%-4s forward: NCHW source %7.3f ms %6.1f TF | record source %7.3f ms %6.1f TF (%.2fx; max diff %.1e);  dgrad: %7.3f | %7.3f ms (%.2fx)"
          % (name, tf0 * 1e3, fl / tf0 / 1e12, tf1 * 1e3, fl / tf1 / 1e12, tf1 / tf0, e, td0 * 1e3, td1 * 1e3, td1 / td0), flush=True)
for name in ([a for a in sys.argv[1:] if a not in DCASES] or ([] if sys.argv[1:] else list(CASES))):
    n, cin, cout, h, w = CASES[name]
    x = torch.randn(n, cin, h, w, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.05
    b = torch.randn(cout, device=dev) * 0.1
    op = K.ConvOp(cin, cout, 3, pad=1)
    fl = 2.0 * n * h * w * cout * cin * 9
    t_old = t(lambda: op.forward(x, wt, b, 0.01, h, w, want_stats=True))
    xr = K.rec_from_nchw(x); wp = K.rconv3_pack(wt)
    pad = torch.zeros((cin // 32, 128), dtype=torch.uint8, device=dev)
    t_new = t(lambda: K.rconv3_forward(xr, wp, b, 0.01, cout, pad_records=pad, want_stats=True))
    t_nos = t(lambda: K.rconv3_forward(xr, wp, b, 0.01, cout, pad_records=pad, want_stats=False))
    t_cv = t(lambda: K.rec_from_nchw(x))
    # the shipped NCHW kernel fed from a record source (record staging in igemm_pipe_kernel, its own epilogue)
    t_xr = t(lambda: op.forward(K.Rec(xr), wt, b, 0.01, h, w, want_stats=True))
    gz = torch.randn(n, cout, h, w, device=dev); gzr = K.rec_from_nchw(gz)
    t_dg = t(lambda: op.dgrad(gz, wt, h, w)); t_dgr = t(lambda: op.dgrad(K.Rec(gzr), wt, h, w))
    print("%-8s igemm_pipe forward: NCHW source %7.3f ms | record source %7.3f ms (%.2fx);  dgrad: %7.3f | %7.3f ms (%.2fx)"
          % (name, t_old * 1e3, t_xr * 1e3, t_xr / t_old, t_dg * 1e3, t_dgr * 1e3, t_dgr / t_dg), flush=True)
    if os.environ.get("PCUDA_RC_DBG") == "1":
        import ctypes
        buf = (ctypes.c_ulonglong * 12)()
        K.L.lib().pcuda_rconv3_debug_clocks(buf)
        for _ in range(10): K.rconv3_forward(xr, wp, b, 0.01, cout, pad_records=pad, want_stats=True)
        K.L.lib().pcuda_rconv3_debug_clocks(buf)
        tot = float(sum(buf)) or 1.0
        if cin == 32 and cout == 32 and os.environ.get("PCUDA_RC_W", "1") != "0":
            names = ["interval top", "MFMA phase", "decode + DMA issue", "values + statistics", "swap / split / stage / stores", "DMA wait", "barrier"] + ["-"] * 5
            print("   two-group kernel, wave 0 of each group (share of time):", " | ".join("%s %.1f%%" % (nm, 100.0 * v / tot) for nm, v in zip(names, buf) if v),
                  " time per workgroup-group and launch: %.0f" % (tot / 10 / 512))
            continue
        names = ["barrier after MFMA", "loop top", "DMA wait", "barrier", "MFMA", "stats store", "epilogue math + LDS stage", "LDS read + global stores",
                 "stats DPP", "barrier", "decode + DMA issue", "-"]
        print("   phases (share of wave 0's time):", " | ".join("%s %.1f%%" % (nm, 100.0 * v / tot) for nm, v in zip(names, buf) if v))
    byts = 4.0 * n * h * w * (cin + cout)
    print("%-8s NCHW kernel %7.3f ms %6.1f TF | record kernel %7.3f ms %6.1f TF %5.2f TB/s (no stats %7.3f ms) | nchw->rec %6.3f ms"
          % (name, t_old * 1e3, fl / t_old / 1e12, t_new * 1e3, fl / t_new / 1e12, byts / t_new / 1e12, t_nos * 1e3, t_cv * 1e3), flush=True)
