#!/bin/bash
# gpurun with retries ONLY for exit code 3 (no box / slot free: nothing ran, nothing charged).  Any other exit code -- the
# command's own result, a refusal, a timeout -- is returned at once: a GPU command is never re-run by this script.
# usage: tools/gpurun_retry.sh <timeout seconds> '<command>'
t="$1"; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ "$rc" -ne 3 ] && exit "$rc"
  sleep 120
done
exit 3
