#!/usr/bin/env python3
"""Average duration of a kernel over its LAST k launches in a rocprofv3 kernel-trace CSV (the timed region of bench.py
is the last `--steps` launches; the 10 cold-start solves and the warm-up precede it)."""
import csv, sys
path, name, k = sys.argv[1], sys.argv[2], int(sys.argv[3])
rows = [r for r in csv.DictReader(open(path)) if name in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[-k:]]
print('%s: %d launches total; last %d: avg %.3f ms, min %.3f, max %.3f' % (name, len(rows), len(d), sum(d) / len(d) / 1e6, min(d) / 1e6, max(d) / 1e6))
