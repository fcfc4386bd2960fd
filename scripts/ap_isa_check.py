#!/usr/bin/env python3
"""ISA rules of the anti-phase convolution kernel (csrc/conv_ap_impl.h), checked on an assembly listing:
 1. v[224:255] -- the landing zone of the input prefetch, in flight across barrier intervals -- appear only inside
    inline-assembly blocks (the compiler must never allocate them);
 2. no AGPR instruction (the 512-entry file is split at compile time: an AGPR use halves the VGPR budget);
 3. no scratch access;
 4. between an inline-assembly ds_read_b128 and the s_waitcnt lgkmcnt that covers it no compiler-generated instruction
    reads or moves the destination registers (reported as a count of v_mov of fragment registers inside MFMA segments).
usage: ap_isa_check.py file.s [kernel-name-substring]"""
import re, sys
path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else 'conv3ap_kernel'
txt = open(path).read().split('\n')
bad = 0
cur = None
inasm = False
hi = re.compile(r'\bv(\d+)\b|\bv\[(\d+):(\d+)\]')
for ln, line in enumerate(txt, 1):
    m = re.match(r'^(_Z\w+):', line)
    if m:
        cur = m.group(1)
    if cur is None or sub not in cur:
        continue
    if 's_endpgm' in line:
        cur = None
        continue
    if '#ASMSTART' in line:
        inasm = True; continue
    if '#ASMEND' in line:
        inasm = False; continue
    code = line.split(';')[0]
    if not code.strip() or code.strip().startswith('.'):
        continue
    if re.search(r'v_accvgpr|\ba\[?\d', code):
        print(f'{path}:{ln}: AGPR use: {code.strip()}'); bad += 1
    if 'scratch_' in code:
        print(f'{path}:{ln}: scratch access: {code.strip()}'); bad += 1
    if not inasm:
        for mm in hi.finditer(code):
            if mm.group(1) is not None:
                lo_ = hi_ = int(mm.group(1))
            else:
                lo_, hi_ = int(mm.group(2)), int(mm.group(3))
            if hi_ >= 224:
                print(f'{path}:{ln}: compiler touches the landing zone: {code.strip()}'); bad += 1
                break
print('violations:', bad)
sys.exit(1 if bad else 0)
