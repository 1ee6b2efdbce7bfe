#!/bin/bash
# A/B the same library under different environments / bench flags: tools/ab_env.sh name "ENV=.. ENV2=.." ...
while [ $# -ge 2 ]; do
  name=$1; envs=$2; shift 2
  out=$(env $envs python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1)
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], 'prof', d['profiled_pass_ms_per_step'], d['kernel_ms_per_step'])"
done
