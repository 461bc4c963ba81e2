#!/bin/bash
# gpurun with retries while no GPU slot is free (exit code 3 = nothing charged, nothing ran)
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
