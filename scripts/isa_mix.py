"""instruction mix of one kernel of an assembly listing, per basic block (loop bodies show as blocks with a back edge).
usage: python scripts/isa_mix.py <file.s> <mangled kernel name prefix> [min_instructions]"""
import re, sys, collections
path, name = sys.argv[1], sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "ds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "ds_write"
    if op.startswith("ds_"): return "ds_other"
    if op.startswith(("global_load", "buffer_load", "flat_load")): return "vmem_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store")): return "vmem_st"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"): return "valu"
    return "other"
blocks, cur, label = [], collections.Counter(), "entry"
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", s):
            blocks.append((label, cur)); cur, label = collections.Counter(), s.split(":")[0]
        continue
    if re.match(r"^\.?[A-Za-z_0-9]+:", s):
        blocks.append((label, cur)); cur, label = collections.Counter(), s.split(":")[0]
        continue
    op = s.split()[0]
    cur[cls(op)] += 1
    if cls(op) == "branch":
        cur["->" + s.split()[-1]] += 0
blocks.append((label, cur))
tot = collections.Counter()
for lb, c in blocks:
    tot.update({k: v for k, v in c.items() if not k.startswith("->")})
    n = sum(v for k, v in c.items() if not k.startswith("->"))
    if n >= minn:
        tg = [k for k in c if k.startswith("->")]
        print("%-12s n=%5d  %s %s" % (lb, n, dict((k, v) for k, v in sorted(c.items()) if not k.startswith("->")), tg))
print("TOTAL", dict(tot))
