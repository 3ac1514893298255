#!/bin/bash
# tools/gpurun_retry.sh <timeout> <log> <command...>: gpurun, retried while the pod has no free GPU slot (exit code 3: nothing charged)
t=$1; log=$2; shift 2
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@" > $log 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
