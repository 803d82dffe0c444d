#!/usr/bin/env python3
"""Audit the gfx950 ISA of k_conv6's loader waves: between an inline-asm global_load and the counted s_waitcnt that
releases it, no instruction may touch the load's destination registers (the compiler does not know they are in flight).
Usage: tools/audit_conv6_isa.py  (compiles the four conv6*_kernels.hip files to assembly with hipcc; CPU only)"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, 'qmri_pnp_recon_poc_amd', 'csrc', f) for f in ('conv6_kernels.hip', 'conv6p_kernels.hip', 'conv6r_kernels.hip', 'conv6s_kernels.hip')]


def regs(tok):
    out = []
    for m in re.finditer(r'v\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out


def sgpr_hazards(lines, name):
    """VALU write of an SGPR (v_readfirstlane; v_readlane = the reload of an SGPR hipcc spilled into a VGPR lane) followed by an inline-asm VMEM
    read of it needs 5 wait states; a VMEM store of more than 8 bytes needs 2 before its data registers are written again."""
    bad = 0
    real = [(i, l.strip()) for i, l in enumerate(lines) if l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.')]
    for k, (i, t) in enumerate(real):
        if not (t.startswith('global_load') or t.startswith('buffer_load') or t.startswith('buffer_store') or t.startswith('global_store')): continue
        need = set()
        for m in re.finditer(r's\[(\d+):(\d+)\]', t): need |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        for m in re.finditer(r'(?<![\w\[])s(\d+)\b', t): need.add(int(m.group(1)))
        if not need: continue
        slots = 0
        for j in range(k - 1, max(k - 12, -1), -1):
            u = real[j][1]
            if slots >= 5: break
            w = re.match(r'v_read(?:first)?lane_b32 s(\d+),', u)
            if w and int(w.group(1)) in need:
                bad += 1
                print('  SGPR HAZARD %s line %d: %s  (written at line %d, %d wait states)' % (name, i + 1, t, real[j][0] + 1, slots))
                break
            n = re.match(r's_nop (\d+)', u)
            slots += (int(n.group(1)) + 1) if n else 1
    in_asm, flag = set(), False
    for i, l in enumerate(lines):
        u = l.strip()
        if u.startswith(';;#ASMSTART'): flag = True
        elif u.startswith(';;#ASMEND'): flag = False
        elif flag: in_asm.add(i)
    for k, (i, t) in enumerate(real):                                 # store data overwritten too early (inline-asm stores: hipcc guards its own)
        if i not in in_asm: continue
        if not (t.startswith('global_store_dwordx4') or t.startswith('buffer_store_dwordx4') or t.startswith('global_store_dwordx3')): continue
        ops = t.split(None, 1)[1].split(',')
        data = set(regs(ops[1] if t.startswith('global_store') else ops[0]))
        slots = 0
        for j in range(k + 1, min(k + 4, len(real))):
            u = real[j][1]
            n = re.match(r's_nop (\d+)', u)
            if n: slots += int(n.group(1)) + 1; continue
            if slots >= 2: break
            if u.startswith('s_') or u.startswith('.LBB'): slots += 1; continue
            dst = u.split(None, 1)[1].split(',')[0] if ' ' in u else ''
            if not (u.startswith('global_store') or u.startswith('buffer_store') or u.startswith('ds_write')) and data & set(regs(dst)):
                bad += 1
                print('  STORE-DATA HAZARD %s line %d: %s   then line %d: %s' % (name, i + 1, t, real[j][0] + 1, u))
                break
            slots += 1
    return bad


def audit(lines, name):
    inasm = False
    pending = []          # list of (line, [regs]) in issue order
    bad = nload = scratch = 0
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(\.LBB\w+):', l)
        if m: labels[m.group(1)] = i
    # a backward branch = a loop: its body is walked a second time with the requests still in flight at the branch (a destination that
    # the NEXT iteration's arithmetic reuses is invisible to a single linear pass -- the prefetch loop of k_conv6 had exactly that)
    order = []
    seen_back = set()
    i = 0
    while i < len(lines):
        order.append(i)
        t = lines[i].strip()
        m = re.match(r's_cbranch_execnz (\.LBB\w+)', t)        # (per-lane loops over elements: `s_andn2 exec ...; s_cbranch_execnz`)
        # (small loops only: the big software-pipelined loops are rotated and branchy -- a linear second walk is not their control flow;
        #  their rotation of register sets is checked by the first pass through the unrolled body)
        if m and m.group(1) in labels and 0 < i - labels[m.group(1)] <= 120 and i not in seen_back:
            seen_back.add(i)
            body = lines[labels[m.group(1)]:i]
            # ... and only loops that request without ever waiting inside (a body with its own counted waits is a pipelined loop, see above)
            if any(('global_load' in b or 'buffer_load' in b) for b in body) and not any('s_waitcnt vmcnt' in b for b in body):
                order += list(range(labels[m.group(1)], i))
        i += 1
    counted = set()
    for i in order:
        l = lines[i]
        t = l.strip()
        if t.startswith(';;#ASMSTART'): inasm = True; continue
        if t.startswith(';;#ASMEND'): inasm = False; continue
        if not t or t.startswith(';') or t.startswith('.'): continue
        first = i not in counted
        counted.add(i)
        if 'scratch_' in t and first: scratch += 1
        if t.startswith('global_load_lds'):
            pending.append((i, []))          # LDS-DMA: a place in the in-order vmcnt queue, no destination registers
            nload += first
            continue
        if inasm and (t.startswith('global_load') or t.startswith('buffer_load')):
            pending.append((i, regs(t.split()[1].rstrip(','))))
            nload += first
            continue
        if t.startswith('global_store') or t.startswith('buffer_store'):
            pending.append((i, []))          # stores take a place in the in-order vmcnt queue (no destination registers)
            continue
        if inasm and t.startswith('s_waitcnt vmcnt'):
            n = int(re.search(r'vmcnt\((\d+)\)', t).group(1))
            pending = pending[-n:] if n > 0 else []
            continue
        if t.startswith('s_branch'):                 # unconditional: what follows in the listing is another path's code, entered with that path's own requests
            pending = []
            continue
        if t.startswith('s_') or t.startswith('.LBB'): continue
        used = set(regs(' '.join(t.split()[1:])))
        for ln, rr in pending:
            if used & set(rr):
                bad += 1
                if bad <= 10: print('  HAZARD %s line %d: %s   (load at line %d)' % (name, i + 1, t, ln + 1))
    sg = sgpr_hazards(lines, name)
    print('%s: %d asm loads, %d hazards, %d SGPR hazards, %d scratch instructions' % (name, nload, bad, sg, scratch))
    return bad + scratch + sg


def main():
    text = []
    with tempfile.TemporaryDirectory() as d:
        for k, src in enumerate(SRCS):
            out = os.path.join(d, 'c6_%d.s' % k)
            subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-w', '-fno-slp-vectorize', '-S', '--cuda-device-only', src, '-o', out])   # (flags of csrc/Makefile)
            text += open(out).read().split('\n')
    total = 0
    cur, name = [], None
    for l in text:
        m = re.match(r'^(_ZN\S*k_conv6\S*):', l)
        if m:
            name, cur = m.group(1)[:40], []
        if name is not None:
            cur.append(l)
            if l.startswith('.Lfunc_end'):
                total += audit(cur, name)
                name = None
    sys.exit(1 if total else 0)


if __name__ == '__main__':
    main()
