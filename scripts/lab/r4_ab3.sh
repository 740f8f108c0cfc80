#!/bin/bash
cd "$(dirname "$0")/../.."
run() {  # label, precision, engine opts, extra args
  out=$(python bench.py --gpus 1 --steps 20 --warmup 3 --precision "$2" --no-cpu-baseline --throughput-mode "" --detail "" --engine-opts "$3" $4 2>/dev/null | tail -1)
  echo "$1 $2 $3 $4 :: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
}
for rep in 1 2; do
  run base     f16gpx3 '{}'
  run mainhi   f16gpx3 '{}' '--main-priority -1'
  run sidehi   f16gpx3 '{"side_priority": -1}'
  run styled   f16gpx3 '{"style_side": true, "style_wg": 0}'
  run base     bf16    '{}'
  run mainhi   bf16    '{}' '--main-priority -1'
  run styled   bf16    '{"style_side": true, "style_wg": 0}'
done
