#!/bin/bash
# usage: tools/gpurun_retry.sh TIMEOUT_S 'command'  -- re-submits only while gpurun answers "no box free" (exit 3: nothing ran, nothing charged)
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  echo "[retry] no box free (attempt $i); sleeping 150 s"
  sleep 150
done
exit 3
