#!/bin/bash
cd "$(dirname "$0")/../.."
run() {  # label, precision, engine opts
  out=$(python bench.py --gpus 1 --steps 20 --warmup 3 --precision "$2" --no-cpu-baseline --throughput-mode "" --detail "" --engine-opts "$3" 2>/dev/null | tail -1)
  echo "$1 $2 $3 :: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
}
for rep in 1 2; do
  run cur      f16gpx3 '{}'
  run c_d      f16gpx3 '{"side_priority": [-1, -1, 0]}'
  run all      f16gpx3 '{"side_priority": [-1, -1, -1]}'
  run c        f16gpx3 '{"side_priority": [-1, 0, 0]}'
  run nopf     f16gpx3 '{"prefetch": false}'
  run c_gp     f16gpx3 '{"side_priority": [-1, 0, -1]}'
done
