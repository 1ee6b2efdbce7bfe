#!/bin/bash
# builds and runs the pivot-block microbenchmark on the GPU box (from the repo root)
#   CFGS="16:1024 8:512"  old-pivot waves : workgroup threads      STAMPS=1  with cycle stamps of the working waves
set -e
mkdir -p gpurun_out
for cfg in ${CFGS:-16:1024 8:512 4:256}; do
  nw=${cfg%%:*}; nt=${cfg##*:}
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=200000 -fno-slp-vectorize -DNWPT=$nw -DBENCH_NT=$nt \
        ${STAMPS:+-DLQP_PIV_STAMPS} ${EXTRA} -o /tmp/piv_bench_$nt tools/microbench/piv_bench.hip 2>/dev/null
  echo "== old pivot: $nw waves; workgroup of $nt threads"
  timeout -k 5 60 /tmp/piv_bench_$nt
done
