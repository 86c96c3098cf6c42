#!/usr/bin/env python3
"""Prints a rocprofv3 `*_kernel_stats.csv` as a short table (kernel, calls, average microseconds, share)."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:44].ljust(46), r["Calls"].rjust(6), "%9.2f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"].rjust(8))
