#!/bin/bash
# gpurun with a retry on "no box or slot free right now" (exit code 3: nothing was charged). Usage: gpurun_retry.sh TIMEOUT 'command'
# (build-container helper; never runs on the GPU box)
T=$1; shift
for i in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
