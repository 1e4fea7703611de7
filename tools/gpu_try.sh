#!/bin/bash
# gpurun with a bounded wait for a free box: retries ONLY the "no box / slot free right now" answer (exit 3, nothing ran,
# nothing charged), at most 20 times, 60 s apart.  Any answer from a box (pass, fail, timeout) ends it.
#   tools/gpu_try.sh <timeout seconds> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
