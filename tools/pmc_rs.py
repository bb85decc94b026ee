#!/usr/bin/env python3
"""Sum the counters of the k_rs2d* launches in a rocprofv3 --pmc csv directory: python tools/pmc_rs.py DIR [name-substring]."""
import csv, glob, sys, collections
d = sys.argv[1]; key = sys.argv[2] if len(sys.argv) > 2 else "k_rs2d"
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if key in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
for k in sorted(tot):
    print("%-32s %14.4g per launch (%d launches)" % (k, tot[k] / n[k], n[k]))
