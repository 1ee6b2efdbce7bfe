#!/bin/bash
# A/B of builds of the library on the headline step, interleaved, two rounds: tools/ab3.sh <other .so> ...   ("" = the default library)
for rep in 1 2; do
  for lib in "" "$@"; do
    if [ -z "$lib" ]; then unset LQP_LIB; else export LQP_LIB=$(realpath $lib); fi
    out=$(python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1)
    echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-default}', d['value'], d['ms_per_step'], d['kernel_ms_per_step'].get('spd_inverse'), d['experiment_1_protocol']['QPs_per_sec_median'])"
  done
done
