#!/usr/bin/env python3
"""Where the GPU hangs (NUMA node, local CPUs) and what the synchronous step costs with the process bound to the local / the
remote socket / not bound (bench.py --sync in child processes)."""
import glob, json, os, subprocess, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = torch.cuda.get_device_properties(0)
bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
path = f"/sys/bus/pci/devices/{bdf}"
node = open(path + "/numa_node").read().strip()
local = open(path + "/local_cpulist").read().strip()
print("GPU", bdf, "numa node", node, "local cpus", local, flush=True)
def expand(s):
    out = []
    for part in s.split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out
loc = set(expand(local))
allc = set(range(os.cpu_count()))
rem = allc - loc
for name, cpus in (("unbound", allc), ("local", loc), ("remote", rem), ("unbound", allc), ("local", loc)):
    if not cpus:
        continue
    code = ("import os,sys,runpy; os.sched_setaffinity(0, %r); sys.argv=['bench.py','--no-cpu-baseline','--no-other-configs','--sync','--steps','40'];"
            "runpy.run_path(%r, run_name='__main__')") % (sorted(cpus), os.path.join(REPO, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=REPO)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "failed", r.stderr[-500:]); continue
    d = json.loads(line[-1])
    print(f"{name:8s} sync step {d['ms_per_step']:.4f} ms = {d['value']:.0f} QPs/s", flush=True)
