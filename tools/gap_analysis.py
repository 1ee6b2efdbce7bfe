#!/usr/bin/env python3
"""Idle time between consecutive GPU dispatches of a rocprofv3 --kernel-trace CSV (sorted by start time).
Usage: tools/gap_analysis.py <kernel_trace.csv> [skip_first_n]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows))
ev = ev[skip:]
gaps = collections.defaultdict(list)
busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    busy += e0 - s0
    gaps[(n0[:40], n1[:40])].append(max(0, s1 - e0))
total = ev[-1][1] - ev[0][0]
print(f"dispatches {len(ev)}  span {total/1e3:.1f} us  busy {busy/1e3:.1f} us  idle {100*(1-busy/total):.1f} %")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print(f"{sum(v)/1e3:9.1f} us total  {sum(v)/len(v)/1e3:7.2f} us avg x{len(v):4d}   {k[0]}  ->  {k[1]}")
