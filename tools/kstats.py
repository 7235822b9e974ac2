#!/usr/bin/env python3
"""Prints the dsph kernels of a rocprofv3 kernel_stats.csv: name, calls, average ns.  Usage: kstats.py <dir>"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsph" in r["Name"]:
            print("%-70s calls %5s avg %9.1f us" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
