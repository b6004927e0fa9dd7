#!/usr/bin/env python3
"""Per-kernel means of the counters in rocprofv3 --pmc CSV output.
usage: tools/pmc_summary.py <dir> [<dir> ...] [--match substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    args = sys.argv[1:]
    match = None
    if '--match' in args:
        i = args.index('--match')
        match = args[i + 1]
        del args[i:i + 2]
    acc = defaultdict(lambda: defaultdict(list))
    for d in args:
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for row in csv.DictReader(open(path)):
                name = row['Kernel_Name']
                if match and match not in name:
                    continue
                short = name.split('(')[0][-70:]
                acc[short][row['Counter_Name']].append(float(row['Counter_Value']))
                acc[short]['_dur_ns'].append(float(row['End_Timestamp']) - float(row['Start_Timestamp']))
    for kern in sorted(acc):
        print(kern)
        for c in sorted(acc[kern]):
            v = acc[kern][c]
            print('   %-32s mean %.4g  (n=%d)' % (c, sum(v) / len(v), len(v)))


if __name__ == '__main__':
    main()
