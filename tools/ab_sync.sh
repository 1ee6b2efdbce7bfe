#!/bin/bash
# A/B of the synchronous (default, reference-semantics) step under the host-wait knobs: tools/ab_sync.sh  (GPU box, repo root)
for rep in 1 2; do
for cfg in "LQP_SYNC_SPLIT=1 LQP_BWD_EARLY=1" "LQP_SYNC_SPLIT=0 LQP_BWD_EARLY=1" "LQP_SYNC_SPLIT=1 LQP_BWD_EARLY=0" "LQP_SYNC_SPLIT=0 LQP_BWD_EARLY=0" "LQP_SYNC_PLAN=0 LQP_SYNC_SPLIT=0"; do
  out=$(env $cfg python bench.py --sync --steps 30 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1)
  echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['value'], d['ms_per_step'])"
done; done
