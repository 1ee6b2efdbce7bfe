#!/usr/bin/env python3
"""Per-kernel summary of one profile set (tools/profile_workload.sh): usage tools/show_profile.py <dir-or-prefix>"""
import csv, json, sys, os
d = sys.argv[1]
f = (lambda n: os.path.join(d, n)) if os.path.isdir(d) else (lambda n: d + n)
rows = list(csv.DictReader(open(f("kernel_stats.csv"))))
sq = json.load(open(f("pmc_sq_raw.json"))); tr = json.load(open(f("traffic.json")))["kernels"]
print(open(f("run.json")).read().strip())
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 10]:
    name = r["Name"].replace("void ", "").split("(")[0]
    if not name.startswith("lqp::"):
        continue
    s = sq.get(name, sq.get("void " + name, {}))
    g = lambda k: s.get(k, {}).get("mean", float("nan"))
    wc = max(g("SQ_WAVE_CYCLES"), 1)
    print(f"{name[:52]:52s} n {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us {r['Percentage'][:5]:>5s}% | HBM {tr.get(name, {}).get('hbm_bytes_per_launch_corrected', 0)/1e6:8.1f} MB"
          f" | MOPS f32 {g('SQ_INSTS_VALU_MFMA_MOPS_F32')/1e6:7.2f}M f64 {g('SQ_INSTS_VALU_MFMA_MOPS_F64')/1e6:6.2f}M | parked {g('SQ_WAIT_ANY')/wc:.2f} issuing {g('SQ_ACTIVE_INST_ANY')/wc:.2f}"
          f" | valu {g('SQ_INSTS_VALU')/1e6:.1f}M lds {g('SQ_INSTS_LDS')/1e6:.1f}M mfma_busy {g('SQ_VALU_MFMA_BUSY_CYCLES')/1e6:.1f}M")
