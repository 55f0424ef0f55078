#!/bin/bash
# Client-side helper (build container): keep asking for a GPU box until the call is accepted.
#   tools/gpu_retry.sh <log> <timeout_s> '<command>'
log=$1; to=$2; shift 2
for i in $(seq 1 60); do
  /usr/local/graft/bin/gpurun --timeout $to -- "$@" > $log 2>&1
  rc=$?
  if ! grep -q "status=transient" $log; then exit $rc; fi
  sleep 45
done
