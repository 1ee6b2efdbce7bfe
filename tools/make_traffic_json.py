#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, separate runs).
Usage: tools/make_traffic_json.py <fetch_dir> <write_dir> <launch_mode> "<command>" > profiles/r01_x_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM: gfx950 tallies
a 128-B request of a wide coalesced read at 64 B)."""
import csv, glob, json, os, sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return acc


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"command": sys.argv[4], "launch_mode": int(sys.argv[3]), "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("lqp::"):
        continue
    f = sum(fetch[k]) / max(len(fetch[k]), 1)
    w = sum(write[k]) / max(len(write[k]), 1)
    out["kernels"][k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "launches": len(fetch[k]) or len(write[k]),
                         "hbm_bytes_per_launch_corrected": int(2 * f * 1024 + w * 1024),
                         "note": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads -> doubled "
                                 "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact"}
print(json.dumps(out, indent=1))
