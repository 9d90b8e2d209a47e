#!/usr/bin/env python3
"""Average duration of the kernels whose name contains one of the given substrings, from a rocprofv3 --kernel-trace --stats csv directory:
    python tools/kernel_us.py <dir> <substring> [...]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(s in r["Name"] for s in sys.argv[2:]):
        print("%-70s calls %5s  avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
