#!/bin/bash
# builds and runs the pivot-block microbenchmark on the GPU box (from the repo root)
set -e
mkdir -p gpurun_out
for cfg in "16 1024" "8 512" "4 256"; do
  set -- $cfg
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=200000 -fno-slp-vectorize -DNWPT=$1 -DBENCH_NT=$2 \
        -o /tmp/piv_bench_$2 tools/microbench/piv_bench.hip 2>/dev/null
  echo "== old pivot: $1 waves; workgroup of $2 threads"
  timeout -k 5 60 /tmp/piv_bench_$2
done
