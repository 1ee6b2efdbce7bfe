#!/bin/bash
# A/B several library builds with the same bench command; prints one line per variant.
# usage: tools/ab_bench.sh name=path.so ...   (path relative to repo root; "base" = default library)
for kv in "$@"; do
  name=${kv%%=*}; lib=${kv#*=}
  if [ "$lib" = "base" ]; then envs=""; else envs="LQP_LIB=$PWD/$lib"; fi
  out=$(env $envs ${EXTRA_ENV} python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1)
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['value'], d['ms_per_step'], 'prof', d['profiled_pass_ms_per_step'], 'loop', d['kernel_ms_per_step']['admm_loop'], 'ach', d['roofline']['achieved'], d['config']['iters'])"
done
