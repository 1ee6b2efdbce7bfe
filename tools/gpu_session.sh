#!/bin/bash
# Run a list of GPU steps back to back; stop at the first step that was killed
# (timeout / signal), never retry.  Usage: tools/gpu_session.sh name1 "cmd1" name2 "cmd2" ...
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
while [ $# -ge 2 ]; do
  name=$1; cmd=$2; shift 2
  echo "=== $name: $cmd" | tee -a gpurun_out/session.log
  start=$(date +%s)
  timeout -k 10 ${STEP_TIMEOUT:-500} bash -c "$cmd" > gpurun_out/$name.log 2>&1
  rc=$?
  echo "=== $name rc=$rc $(( $(date +%s) - start ))s" | tee -a gpurun_out/session.log
  if [ $rc -ge 124 ]; then echo "step $name killed (rc=$rc): stopping" | tee -a gpurun_out/session.log; exit $rc; fi
done
exit 0
