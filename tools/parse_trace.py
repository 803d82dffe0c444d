#!/usr/bin/env python3
"""Group a rocprofv3 kernel_trace.csv by (kernel, grid) and print mean/min durations in microseconds."""
import collections
import csv
import glob
import sys

paths = sys.argv[1:] or glob.glob("gpurun_out/**/*kernel_trace.csv", recursive=True)
for p in paths:
    g = collections.defaultdict(list)
    for r in csv.DictReader(open(p)):
        name = r["Kernel_Name"]
        short = name.split("(")[0].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:40]
        key = (short, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
        g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
    print(p)
    tot = sum(sum(v) for v in g.values())
    for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {k[0]:40s} grid=({k[1]},{k[2]},{k[3]}) vgpr={k[4]}+{k[5]} lds={k[6]:>6s} n={len(v):4d} avg={sum(v)/len(v):8.1f} min={min(v):8.1f} share={100*sum(v)/tot:5.1f}%")
