#!/usr/bin/env python3
"""Kernel timeline of ONE PnP-ADMM iteration from a rocprofv3 --kernel-trace of bench.py (the last complete iteration of the trace):
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/it -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-roofline
   python tools/iter_times.py gpurun_out/it
Non-conv launches one by one (duration, idle time before), convolutions summed."""
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'k_unnormalise_dual' in r[2] or 'k_dual_fwd_h' in r[2]]     # (the iteration's last launch; round 4: the fused kernel)
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
lo, hi = marks[which] + 1, marks[which + 1] + 1
it = rows[lo:hi]
short = lambda n: n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:44]
print(f'iteration: {len(it)} launches, {(it[-1][1] - rows[lo - 1][1]) / 1e3:.1f} us from the previous iteration\'s last kernel end to this one\'s')
prev = rows[lo - 1][1]
conv = [0, 0.0, 0.0]
for s, e, n in it:
    name = short(n)
    idle = (s - prev) / 1e3
    if name.startswith('k_conv') or name.startswith('k_act'):
        conv[0] += 1; conv[1] += (e - s) / 1e3; conv[2] += max(idle, 0.0)
    else:
        if conv[0]:
            print(f'  [{conv[0]} network launches: {conv[1]:.1f} us busy, {conv[2]:.1f} us idle between them]'); conv = [0, 0.0, 0.0]
        print(f'  {name:40s} {(e - s) / 1e3:8.2f} us   idle before {idle:6.2f}')
    prev = max(prev, e)
