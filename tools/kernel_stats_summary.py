#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats.csv: name, calls, total and average duration (us)."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
total = sum(float(r['TotalDurationNs']) for r in rows)
print(f'total GPU time {total / 1e6:.3f} ms in {sum(int(r["Calls"]) for r in rows)} launches')
for r in rows[:n]:
    print(f"{r['Name'][:96]:96s} {int(r['Calls']):6d} tot {float(r['TotalDurationNs']) / 1e3:10.1f} us  avg {float(r['AverageNs']) / 1e3:8.1f}")
