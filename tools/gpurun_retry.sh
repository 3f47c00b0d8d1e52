#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (rc 3: nothing charged).  usage: tools/gpurun_retry.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 60
done
exit 3
