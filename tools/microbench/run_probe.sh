#!/bin/bash
hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -pragma-unroll-threshold=200000 -fno-slp-vectorize -DLQP_PIV_DEBUG_STOP -o /tmp/pdbg tools/microbench/piv_debug.hip 2>/dev/null && timeout -k 5 30 /tmp/pdbg
