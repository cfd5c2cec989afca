#!/bin/bash
# Waits for a free GPU slot (gpurun exit code 3 = none free, nothing charged) and then runs ONE gpurun call.
# Usage: tools/gpu_when_free.sh TIMEOUT 'command'
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
