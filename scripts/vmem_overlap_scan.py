"""List the vector-memory loads of a directory of AMDGPU assembly files (hipcc -S --cuda-device-only) whose DESTINATION
registers overlap their own ADDRESS registers (`global_load_dwordx4 v[46:49], v[46:47], off`).  Legal for the compiler;
on a GPU shared by two processes such loads returned wrong data in this project's direct kernels (common.h, VMEM address
rule; profiles/r02_two_process_determinism.txt).  usage: python scripts/vmem_overlap_scan.py <dir with *.s> [kernel-name substring ...]"""
import re, subprocess, sys, glob
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m: return int(m.group(1)), int(m.group(2))
    m = re.match(r"v(\d+)$", tok)
    if m: return int(m.group(1)), int(m.group(1))
    return None
def scan(directory, only=()):
    """-> list of (file, demangled kernel, count, first example)"""
    out = []
    for f in sorted(glob.glob(directory + "/*.s")):
        txt = open(f).read()
        for m in re.finditer(r"^(_Z[^\n:]+):.*?s_endpgm", txt, re.S | re.M):
            n, ex = 0, []
            for l in m.group(0).split("\n"):
                mm = re.match(r"\s*((?:global|flat|buffer|scratch)_load_\w+)\s+([^,]+),\s*([^,]+)", l)
                if not mm:
                    continue
                d, a = regs(mm.group(2).strip()), regs(mm.group(3).strip())
                if d and a and not (d[1] < a[0] or d[0] > a[1]):
                    n += 1; ex.append(l.strip())
            dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            if only and not any(o in dem for o in only):
                continue
            out.append((f.split("/")[-1], dem, n, ex[0] if ex else ""))
    return out


if __name__ == "__main__":
    tot = 0
    for f, dem, n, ex in scan(sys.argv[1], sys.argv[2:]):
        if n:
            tot += n
            print("%-24s %3d  %s   e.g. %s" % (f, n, dem[:90], ex))
    print("total", tot)
    sys.exit(0)
