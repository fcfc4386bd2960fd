"""schedule of conv3rs_kernel's main-loop steps as one character per instruction (M MFMA, r / w LDS read / write, L / S global
load / store, v VALU, . SALU, [..] waits): python3 scripts/rs_isa_view.py [mangled-name-fragment]"""
import re, sys, collections
s = open('pointcloududa_amd/csrc/build/isa/conv_rs.s').read()
frag = sys.argv[1] if len(sys.argv) > 1 else 'ILi1ELb0ELb1E'
m = re.search(r'^(\S*conv3rs_kernel' + frag + r'\S*):', s, re.M)
name = m.group(1)
i = s.index(name + ':'); j = s.index('.end_amdhsa_kernel', i)
body = s[i:j].split('\n')
meta = s[j - 4000:j]
out = []
for l in body:
    t = l.strip()
    if re.match(r'\.LBB\d+_\d+:', t): out.append('\n' + t.split(':')[0] + ': ')
    if not l.startswith('\t') or not t or t.startswith('.') or t.startswith(';'): continue
    op = t.split()[0]
    if op.startswith('v_mfma'): ch = 'M'
    elif op.startswith('ds_read'): ch = 'r'
    elif op.startswith('ds_write'): ch = 'w'
    elif op.startswith('buffer_load') or op.startswith('global_load'): ch = 'L'
    elif op.startswith('buffer_store') or op.startswith('global_store'): ch = 'S'
    elif op.startswith('s_waitcnt'):
        a = re.search(r'vmcnt\((\d+)\)', t); b = re.search(r'lgkmcnt\((\d+)\)', t)
        ch = '[' + ('v%s' % a.group(1) if a else '') + ('l%s' % b.group(1) if b else '') + ']'
    elif op.startswith('s_cbranch') or op.startswith('s_branch'): ch = 'B'
    elif op.startswith('s_'): ch = '.'
    elif op.startswith('v_accvgpr'): ch = 'a'
    elif op.startswith('v_'): ch = 'v'
    else: ch = '?'
    out.append(ch)
txt = ''.join(out)
for blk in txt.split('\n'):
    if blk.count('M') >= 50: print(blk[:2600]); print()
ins = [l.strip().split()[0] for l in body if l.startswith('\t') and l.strip() and not l.strip().startswith('.') and not l.strip().startswith(';')]
print(len(ins), collections.Counter(ins).most_common(12))
