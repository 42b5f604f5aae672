#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (status=transient: nothing ran, nothing charged).
# usage: tools/gpu_retry.sh TIMEOUT 'command'
for i in $(seq 1 20); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "$2" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
  echo "$out"; exit 0
done
echo "$out"; exit 3
