#!/bin/bash
# rocprofv3 material of ONE workload of tools/profile_workload.py (run on the GPU box from the repo root):
#   kernel trace + stats, FETCH_SIZE and WRITE_SIZE (separate passes), the SQ / matrix-instruction counters.
# usage: tools/profile_workload.sh <tag> <workload> [steps]      -> gpurun_out/<tag>_<workload>/{run.json, kernel_stats.csv, traffic.json, pmc_sq_raw.json}
# then copy those four to profiles/<tag>_<workload>_*.  The program stands directly after `--` (no env / shell hop).
root=$(pwd); tag=${1:?tag}; wl=${2:?workload}; steps=${3:-5}; warm=3
out=$root/gpurun_out/${tag}_${wl}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P="python3 $root/tools/profile_workload.py $wl --steps $steps --warmup $warm"
timeout -k 10 300 $P > $out/run.json 2>$out/run.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $P > $out/kt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $P > $out/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $P > $out/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $out/sq -- $P > $out/sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/sq2 -- $P > $out/sq2.log 2>&1 || true
cd $root
python3 tools/make_traffic_json.py $out/fetch $out/write 3 "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv (separate passes) -- python3 tools/profile_workload.py $wl --steps $steps --warmup $warm" > $out/traffic.json
python3 tools/pmc_summary.py $out/sq $out/sq2 > $out/pmc_sq_raw.json
cp $(ls $out/kt/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
rm -rf $out/kt $out/fetch $out/write $out/sq $out/sq2
cat $out/run.json
