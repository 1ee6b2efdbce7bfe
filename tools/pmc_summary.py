#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs per kernel (average per dispatch).
Usage: tools/pmc_summary.py <dir-with-*_counter_collection.csv> [...]"""
import csv, glob, json, os, sys
from collections import defaultdict

out = {}
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, vals in cs.items():
                out.setdefault(k, {})[c] = {"mean": sum(vals) / len(vals), "n": len(vals)}
print(json.dumps(out, indent=1))
