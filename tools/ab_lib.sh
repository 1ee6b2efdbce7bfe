#!/bin/bash
# A/B of two builds of the library on the headline step, interleaved, three rounds: tools/ab_lib.sh <other .so>
other=$(realpath $1)
for rep in 1 2 3; do
  for lib in "" "$other"; do
    if [ -z "$lib" ]; then unset LQP_LIB; else export LQP_LIB=$lib; fi
    out=$(python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1)
    echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-default}', d['value'], d['ms_per_step'], d['kernel_ms_per_step'].get('spd_inverse'), d['experiment_1_protocol']['QPs_per_sec_median'])"
  done
done
