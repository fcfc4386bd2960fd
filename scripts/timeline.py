"""union / per-queue busy time of the kernels of the last benchmark steps from a rocprofv3 --kernel-trace csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Queue_Id", "?"))
             for r in rows), key=lambda e: e[0])
# step boundaries: the first kernel of a step is the gradient memset/zero of the segmenter (adam comes last): use adam_kernel ends
ends = [e[1] for e in ev if e[2].startswith("adam_kernel") or "adam" in e[2]]
if len(ends) < nsteps + 1:
    print("not enough steps", len(ends)); sys.exit()
t0, t1 = ends[-nsteps - 1], ends[-1]
sel = [e for e in ev if e[0] >= t0 and e[1] <= t1 + 10_000_000]
sel = [e for e in sel if e[0] < t1]
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in sel)
print("window %.2f ms/step   union busy %.2f   sum of kernel time %.2f   idle %.2f   kernels/step %d" % (
    (t1 - t0) / 1e6 / nsteps, busy / 1e6 / nsteps, tot / 1e6 / nsteps, (t1 - t0 - busy) / 1e6 / nsteps, len(sel) // nsteps))
perq = collections.defaultdict(float)
for s, e, _, q in sel:
    perq[q] += e - s
for q, v in sorted(perq.items(), key=lambda kv: -kv[1]):
    print("  queue %s  %.2f ms/step" % (q, v / 1e6 / nsteps))
pern = collections.defaultdict(lambda: [0.0, 0])
for s, e, n, _ in sel:
    pern[n][0] += e - s; pern[n][1] += 1
for n, (v, c) in sorted(pern.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-60s %.2f ms/step  n=%d" % (n, v / 1e6 / nsteps, c // nsteps))
# largest idle gaps of the busiest queue (the caller's stream) and what surrounds them
mainq = max(perq.items(), key=lambda kv: kv[1])[0]
mq = [e for e in sel if e[3] == mainq]
gaps = []
for a, b in zip(mq, mq[1:]):
    if b[0] - a[1] > 100_000:
        gaps.append((b[0] - a[1], a, b))
print("main-queue gaps > 0.1 ms: total %.2f ms/step" % (sum(g[0] for g in gaps) / 1e6 / nsteps))
for d, a, b in sorted(gaps, key=lambda g: -g[0])[:12]:
    others = collections.Counter(e[2] for e in sel if e[3] != mainq and e[0] < b[0] and e[1] > a[1])
    print("  %.2f ms at +%.1f ms  after %-28s before %-28s | other queues: %s" % (
        d / 1e6, (a[1] - t0) / 1e6, a[2][:28], b[2][:28], ", ".join("%s x%d" % (k[:22], v) for k, v in others.most_common(3))))
