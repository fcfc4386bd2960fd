"""summarise a PCUDA_PROF_DUMP csv: per shape tag total ms, TFLOP/s (or TB/s for the pointwise family), launches, and --
for the convolution launches -- the achieved ALGORITHMIC GB/s (input read once + output written once, fp32: what the
layer's HBM roofline is measured against)"""
import collections, csv, re, sys
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["family"], r["tag"])
    ms, w = float(r["ms"]), float(r["work"])
    a = rows.setdefault(k, [0.0, 0.0, 0])
    a[0] += ms; a[1] += w; a[2] += 1


def alg_bytes(tag):
    """fp32 bytes of one launch: reduction-side tensor + row-side tensor at the launch's logical size"""
    m = re.match(r"igemm n(\d+) red(\d+) rows(\d+) (\d+)x(\d+) taps(\d+) step(\d+) up(\d)", tag)
    if m:
        n, red, rows_, lh, lw, taps, step, up = (int(v) for v in m.groups())
        inpix = lh * lw * step * step / (4 if up else 1)          # stride-2 forward reads 4 input pixels per output pixel
        return 4.0 * n * (red * inpix + rows_ * lh * lw)
    m = re.match(r"wgrad n(\d+) cin(\d+) cout(\d+) (\d+)x(\d+) k(\d+) s(\d+)", tag)
    if m:
        n, cin, cout, oh, ow, k, s = (int(v) for v in m.groups())
        return 4.0 * n * (cin * oh * ow * s * s + cout * oh * ow)
    return None


tot = sum(a[0] for a in rows.values())
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for (fam, tag), (ms, w, n) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:top]:
    b = alg_bytes(tag)
    gbs = "%6.0f GB/s" % (b * n / ms / 1e6) if (b and ms) else " " * 11
    print("%7.2f ms %5.1f%% n=%3d %7.1f T/s %s  fam%s %s" % (ms, 100 * ms / tot, n, w / ms / 1e9 if ms else 0, gbs, fam, tag))
print("total %.2f ms" % tot)
