#!/bin/bash
cd "$(dirname "$0")/../.."
run() {  # label, precision, engine opts, extra args
  out=$(python bench.py --gpus 1 --steps 20 --warmup 3 --precision "$2" --no-cpu-baseline --throughput-mode "" --detail "" --engine-opts "$3" $4 2>/dev/null | tail -1)
  echo "$1 $2 $3 $4 :: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
}
for rep in 1 2; do
  run base     f16gpx3 '{}'
  run m_gp     f16gpx3 '{"side_priority": [0, 0, -1]}' '--main-priority -1'
  run m_d_gp   f16gpx3 '{"side_priority": [0, -1, -1]}' '--main-priority -1'
  run m_d      f16gpx3 '{"side_priority": [0, -1, 0]}' '--main-priority -1'
  run d_gp     f16gpx3 '{"side_priority": [0, -1, -1]}'
  run gp       f16gpx3 '{"side_priority": [0, 0, -1]}'
  run m_d      bf16    '{"side_priority": [0, -1, 0]}' '--main-priority -1'
  run m_c      bf16    '{"side_priority": [-1, 0, 0]}' '--main-priority -1'
done
