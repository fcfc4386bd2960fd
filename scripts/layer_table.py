"""summarise a PCUDA_PROF_DUMP csv: per shape tag total ms, TFLOP/s (or TB/s), launches"""
import collections, csv, sys
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["family"], r["tag"])
    ms, w = float(r["ms"]), float(r["work"])
    a = rows.setdefault(k, [0.0, 0.0, 0])
    a[0] += ms; a[1] += w; a[2] += 1
tot = sum(a[0] for a in rows.values())
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for (fam, tag), (ms, w, n) in sorted(rows.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%7.2f ms %5.1f%% n=%3d %7.1f T/s  fam%s %s" % (ms, 100 * ms / tot, n, w / ms / 1e9 if ms else 0, fam, tag))
print("total %.2f ms" % tot)
