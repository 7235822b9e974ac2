#!/bin/bash
# usage (here): tools/gr.sh <timeout-s> <out-file> '<remote command>' -- gpurun with retries while all GPU slots of the pod are busy
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$3" > "$2" 2>&1
  if ! grep -q "status=transient" "$2"; then break; fi
  sleep 45
done
tail -2 "$2"
