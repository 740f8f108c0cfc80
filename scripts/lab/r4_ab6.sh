#!/bin/bash
cd "$(dirname "$0")/../.."
run() {  # label, precision, engine opts, extra args
  out=$(python bench.py --gpus 1 --steps 20 --warmup 3 --precision "$2" --no-cpu-baseline --throughput-mode "" --detail "" --engine-opts "$3" $4 2>/dev/null | tail -1)
  echo "$1 $2 $3 $4 :: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
}
for rep in 1 2; do
  run das0     bf16    '{"disc_after_style": false}'
  run das1     bf16    '{"disc_after_style": true}'
  run das0     f16gpx3 '{"disc_after_style": false}'
  run das1     f16gpx3 '{"disc_after_style": true}'
done
