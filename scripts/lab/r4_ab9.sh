#!/bin/bash
# same-box comparison of two source trees (HEAD vs a worktree of an earlier commit under _old/)
cd "$(dirname "$0")/../.."
run() {  # label, bench path, precision, engine opts
  out=$(python $2 --gpus 1 --steps 20 --warmup 3 --precision "$3" --no-cpu-baseline --throughput-mode "" --detail "" --engine-opts "$4" 2>/dev/null | tail -1)
  echo "$1 $3 $4 :: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
}
for rep in 1 2; do
  run old  _old/bench.py f16gpx3 '{}'
  run new  bench.py      f16gpx3 '{}'
  run new_nopf bench.py  f16gpx3 '{"prefetch": false}'
  run old  _old/bench.py bf16 '{}'
  run new  bench.py      bf16 '{}'
done
