#!/bin/bash
# Round 4 schedule A/B (2): style backward beside the wide weight-gradient launch, stream priorities.
cd "$(dirname "$0")/../.."
run() {  # label, precision, engine opts, extra args
  out=$(python bench.py --gpus 1 --steps 20 --warmup 3 --precision "$2" --no-cpu-baseline --throughput-mode "" --detail "" --engine-opts "$3" $4 2>/dev/null | tail -1)
  echo "$1 $2 $3 $4 :: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
}
for rep in 1 2; do
  run nostyle  f16gpx3 '{"style_side": false}'
  run style32  f16gpx3 '{"style_side": true, "style_wg": 32}'
  run style64  f16gpx3 '{"style_side": true, "style_wg": 64}'
  run mainhi   f16gpx3 '{}' '--main-priority -1'
  run nostyle  bf16    '{"style_side": false}'
  run style32  bf16    '{"style_side": true}'
  run mainhi   bf16    '{}' '--main-priority -1'
done
run style_early f16gpx3 '{"style_early": true}'
run style_early bf16 '{"style_early": true}'
run tn_early bf16 '{"style_side": false, "tn_early": true}'
