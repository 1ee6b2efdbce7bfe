#!/bin/bash
# kernel trace of a few bench steps: which launches (ours, torch's, the runtime's copy / fill kernels) one step consists of
# usage (GPU box, repo root): bash tools/gpu_trace_step.sh [extra bench.py flags, e.g. --sync]  ->  gpurun_out/trace_step.txt
root=$(pwd); out=$root/gpurun_out/trace; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/kt -- python3 $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs "$@" > $out/kt.log 2>&1 || exit 1
cd $root
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/trace/kt/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K ' + r['Kernel_Name'][:70]))
for f in glob.glob('gpurun_out/trace/kt/*/*memory_copy_trace.csv'):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', ''))))
rows.sort()
with open('gpurun_out/trace_step.txt', 'w') as o:
    prev = None
    for s, e, name in rows[-120:]:
        o.write(f"{(s - prev) / 1e3 if prev else 0:9.2f} gap  {(e - s) / 1e3:9.2f} us  {name}\n")
        prev = e
PY
rm -rf $out/kt
