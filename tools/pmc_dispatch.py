#!/usr/bin/env python3
"""Counters of every dispatch of the kernels matching a substring, one line per dispatch.
usage: tools/pmc_dispatch.py <dir> <substring>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d, match = sys.argv[1], sys.argv[2]
per = defaultdict(dict)
for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for row in csv.DictReader(open(path)):
        if match not in row['Kernel_Name']:
            continue
        k = int(row['Dispatch_Id'])
        per[k][row['Counter_Name']] = float(row['Counter_Value'])
        per[k]['dur_us'] = (float(row['End_Timestamp']) - float(row['Start_Timestamp'])) / 1e3
ks = sorted(per)
cols = sorted(per[ks[0]])
print(' '.join('%14s' % c[-14:] for c in ['id'] + cols))
for k in ks:
    print(' '.join('%14.4g' % v for v in [k] + [per[k].get(c, float('nan')) for c in cols]))
