#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc csv (counter_collection.csv): mean counter value per kernel name.
Usage: python tools/pmc_summary.py <dir-or-csv> [name-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for f in files:
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if flt and flt not in name:
                continue
            acc[name[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name, ctrs in acc.items():
    print(name)
    for c, v in sorted(ctrs.items()):
        print(f"    {c:34s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
