#!/bin/bash
cd "$(dirname "$0")/../.."
run() {  # label, precision, engine opts, extra args
  out=$(python bench.py --gpus 1 --steps 20 --warmup 3 --precision "$2" --no-cpu-baseline --throughput-mode "" --detail "" --engine-opts "$3" $4 2>/dev/null | tail -1)
  echo "$1 $2 $3 $4 :: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
}
for rep in 1 2; do
  run nopf     bf16    '{"prefetch": false}'
  run pf       bf16    '{"prefetch": true}'
  run pf_st2   bf16    '{"prefetch": true, "style_side": 2}'
  run nopf     f16gpx3 '{"prefetch": false}'
  run pf       f16gpx3 '{"prefetch": true}'
  run pf_st2   f16gpx3 '{"prefetch": true, "style_side": 2}'
done
