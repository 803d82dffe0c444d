#!/usr/bin/env python3
"""Basic blocks of one kernel in a hipcc -S listing: instruction counts per block (MFMA / vector ALU / scalar / vector memory / LDS).
Usage: tools/isa_blocks.py file.s <substring of the kernel's symbol> [--dump <block label>]"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
blocks, cur = [], ['<entry>']
for l in s[start + 1:end + 1]:
    t = l.strip()
    if re.match(r'^\.LBB\d+_\d+:', t): blocks.append(cur); cur = [t.split(':')[0]]
    elif t and not t.startswith((';', '.')): cur.append(t.split(';')[0].strip())
blocks.append(cur)
if '--dump' in sys.argv:
    lab = sys.argv[sys.argv.index('--dump') + 1]
    for b in blocks:
        if b[0] == lab: print('\n'.join(b))
    sys.exit(0)
for b in blocks:
    ins = b[1:]
    c = lambda f: sum(1 for x in ins if f(x))
    print('%-12s n %4d  mfma %3d  valu %4d  salu %4d  vmem %3d  lds %3d  branch %s' % (
        b[0], len(ins), c(lambda x: 'v_mfma' in x), c(lambda x: x.startswith('v_') and 'mfma' not in x), c(lambda x: x.startswith('s_')),
        c(lambda x: x.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))), c(lambda x: x.startswith('ds_')),
        ' '.join(x.split()[-1] for x in ins if x.startswith('s_cbranch') or x.startswith('s_branch'))))
for l in s[end:end + 400]:
    if any(k in l for k in ('.vgpr_count', '.sgpr_count', 'vgpr_spill', 'private_segment_fixed_size', 'agpr_count', 'next_free_vgpr', 'accum_offset')): print(l.strip())
