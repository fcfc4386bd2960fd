"""The row-streaming kernel (csrc/conv_rs.hip) has ONE wave per SIMD: nothing but its own instruction order puts the conversion,
the epilogue and the loads into the shadow of the MFMAs (scheduling fences pin one piece of that work behind every MFMA; the
sched_group_barrier solver left all of it outside the MFMA sequence).  A compiler that reorders across the fences, spills, or
clusters the MFMAs again would cost the kernel its overlap silently.  For every shipped instantiation (DBG = 0) check in the
assembly: no scratch in the variants the networks launch, 54 MFMAs per step copy, and inside the main loop no long run of vector
instructions without an MFMA and no long run of MFMAs.   usage: python3 scripts/rs_isa_check.py build/isa/conv_rs.s"""
import re
import sys


def kernels(txt):
    for m in re.finditer(r"^(_ZN\S*conv3rs_kernel\S*):[^\n]*\n(.*?)\.end_amdhsa_kernel", txt, re.S | re.M):
        yield m.group(1), m.group(2)


def classify(line):
    t = line.strip()
    if not line.startswith("\t") or not t or t.startswith((".", ";")):
        return None
    op = t.split()[0]
    if op.startswith("v_mfma"):
        return "M"
    if op.startswith("v_"):
        return "v"
    return "o"


def check(path):
    txt = open(path).read()
    bad = []
    n = 0
    for name, body in kernels(txt):
        tp = re.search(r"conv3rs_kernelILi(\d)ELb(\d)ELb(\d)ELi(\d+)E", name)
        if not tp or tp.group(4) != "0":
            continue
        n += 1
        stats, acc = int(tp.group(1)), int(tp.group(2))
        scratch = int(re.search(r"private_segment_fixed_size (\d+)", body).group(1))
        if not acc and scratch:
            bad.append("%s: %d bytes of scratch" % (name, scratch))
        seq = [c for c in (classify(l) for l in body.split("\n")) if c]
        nm = seq.count("M")
        if nm != 54 * 8:
            bad.append("%s: %d MFMAs, expected %d (first step + four loop steps + three remainder steps)" % (name, nm, 54 * 8))
        # between the first and the last MFMA: runs of vector instructions / of MFMAs
        first, last = seq.index("M"), len(seq) - 1 - seq[::-1].index("M")
        run_v = run_m = best_v = best_m = 0
        for c in seq[first:last + 1]:
            if c == "M":
                run_m += 1; best_m = max(best_m, run_m); run_v = 0
            elif c == "v":
                run_v += 1; best_v = max(best_v, run_v); run_m = 0
            # (scalar / memory instructions break neither run)
        if best_v > 56:
            bad.append("%s: %d vector instructions in a row inside the MFMA sequence" % (name, best_v))
        if best_m > 30:      # (the last 19 MFMAs of a step carry no pinned work)
            bad.append("%s: %d MFMAs in a row" % (name, best_m))
    if n < 6:
        bad.append("expected at least six shipped instantiations, found %d" % n)
    return bad


if __name__ == "__main__":
    problems = check(sys.argv[1])
    for p in problems:
        print(p)
    print("conv3rs_kernel: %s" % ("FAILED" if problems else "ok"))
    sys.exit(1 if problems else 0)
