#!/bin/bash
# regenerates the material of profiles/<tag>_* under gpurun_out/<tag>/ (run on the GPU box from the repo root)
# usage: tools/profile_session.sh r02_a ; then copy bench*.log, bench_kernel_stats.csv, traffic.json, pmc_sq_raw.json to profiles/<tag>_*
root=$(pwd); tag=${1:-r02_x}; out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs"
timeout -k 10 400 python3 $root/bench.py > $out/bench.json.log 2>$out/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $out/kt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > $out/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > $out/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $out/sq -- $B > $out/sq.log 2>&1 || exit 1
cd $root
python3 tools/make_traffic_json.py $out/fetch $out/write 3 "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv (separate passes) -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs" > $out/traffic.json
python3 tools/pmc_summary.py $out/sq > $out/pmc_sq_raw.json
cp $(ls $out/kt/*/*kernel_stats.csv | head -1) $out/bench_kernel_stats.csv
rm -rf $out/kt $out/fetch $out/write $out/sq
ls -la $out
