#!/bin/bash
# hardware-queue count with the round-4 stream set (default stream + the agent's update stream + critic + discriminator [+ penalty value path])
cd "$(dirname "$0")/../.."
run() {  # label, precision, queues
  out=$(GPU_MAX_HW_QUEUES=$3 python bench.py --gpus 1 --steps 20 --warmup 3 --precision "$2" --no-cpu-baseline --throughput-mode "" --detail "" 2>/dev/null | tail -1)
  echo "$1 $2 q=$3 :: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
}
for rep in 1 2; do
  for q in 4 5 6 8; do run hwq f16gpx3 $q; done
  for q in 4 5 6; do run hwq bf16 $q; done
done
