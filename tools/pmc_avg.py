#!/usr/bin/env python3
"""Average every counter of a rocprofv3 counter_collection CSV over the fg_kernel dispatches."""
import collections
import csv
import sys

acc = collections.defaultdict(list)
with open(sys.argv[1], newline="") as fh:
    for r in csv.DictReader(fh):
        if "fg_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("%-32s avg %18.1f  n=%d" % (k, sum(v) / len(v), len(v)))
