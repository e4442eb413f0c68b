#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel.
   python tools/pmc_summary.py gpurun_out/pmc1 [gpurun_out/pmc2 ...] [--filter fc_]"""
import csv
import glob
import sys
from collections import defaultdict

flt = 'fc_'
dirs = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == '--filter':
        flt = args.pop(0)
    else:
        dirs.append(a)
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name']
            if flt not in name:
                continue
            short = name.split('(')[0].replace('void ', '')
            acc[short][row['Counter_Name']].append(float(row['Counter_Value']))
for k, ctrs in acc.items():
    print(k)
    for c, v in sorted(ctrs.items()):
        print(f'   {c:32s} mean {sum(v) / len(v):16.1f}  (n={len(v)})')
