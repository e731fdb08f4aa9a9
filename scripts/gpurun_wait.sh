#!/bin/bash
# gpurun, retried only while the pool reports "no box or slot free" (exit code 3: nothing ran, nothing was charged)
# usage: scripts/gpurun_wait.sh TIMEOUT 'command'
for try in $(seq 1 15); do
  /usr/local/graft/bin/gpurun --timeout $1 -- "$2"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
