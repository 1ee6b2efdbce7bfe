#!/bin/bash
# tools/ab_flags.sh name "bench flags" ...
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  out=$(python bench.py --steps 10 --warmup 3 --no-cpu-baseline $flags 2>/dev/null | tail -1)
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], 'prof', d['profiled_pass_ms_per_step'])"
done
