#!/bin/bash
# per-kernel average durations of one short bench run (rocprofv3 --kernel-trace --stats); run on the GPU box
# usage: tools/kernel_times.sh [tag]
tag=${1:-kt}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $root/gpurun_out/prof_$tag.log 2>&1
cd $root
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
for row in list(csv.DictReader(open(sys.argv[1])))[:12]:
    name = row["Name"].replace("void ", "").split("(")[0][:60]
    print(f"{name:60s} calls {row['Calls']:>5s} avg_us {float(row['AverageNs'])/1e3:9.1f} pct {row['Percentage']}")
PY
grep -o '"value": [0-9.]*' gpurun_out/prof_$tag.log | head -1
