#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run (either the *_kernel_trace.csv or the rocpd *_results.db)."""
import csv, re, sqlite3, sys
from collections import defaultdict


def short(n):
    n = re.sub(r'^void\s+', '', n)
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n[:64]


def main(p):
    acc = defaultdict(lambda: [0, 0.0])
    if p.endswith('.db'):
        c = sqlite3.connect(p)
        for n, s, e in c.execute('select name, start, end from kernels'):
            a = acc[short(n)]; a[0] += 1; a[1] += e - s
    else:
        for r in csv.DictReader(open(p)):
            a = acc[short(r['Kernel_Name'])]; a[0] += 1; a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    tot = sum(v[1] for v in acc.values())
    print(f"{'kernel':64s} {'calls':>6s} {'avg us':>9s} {'total ms':>9s} {'%':>6s}")
    for n, (k, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print(f"{n:64s} {k:6d} {s / k / 1e3:9.2f} {s / 1e6:9.2f} {100 * s / tot:6.1f}")


if __name__ == '__main__':
    main(sys.argv[1])
