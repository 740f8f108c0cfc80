#!/bin/bash
# Round 4 schedule A/B on one box: cross-step discriminator head (xstep) and the penalty value path on its own stream (gp_stream).
# usage (through gpurun): bash scripts/lab/r4_ab.sh > gpurun_out/r4_ab.log 2>&1
cd "$(dirname "$0")/../.."
run() {  # label, precision, engine opts
  out=$(python bench.py --gpus 1 --steps 20 --warmup 3 --precision "$2" --no-cpu-baseline --throughput-mode "" --detail "" --engine-opts "$3" 2>/dev/null | tail -1)
  echo "$1 $2 $3 :: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
}
for rep in 1 2; do
  run base   f16gpx3 '{"xstep": false, "gp_stream": false}'
  run gps    f16gpx3 '{"xstep": false, "gp_stream": true}'
  run xs     f16gpx3 '{"xstep": true, "gp_stream": false}'
  run both   f16gpx3 '{"xstep": true, "gp_stream": true}'
  run base   bf16    '{"xstep": false}'
  run xs     bf16    '{"xstep": true}'
done
GPU_MAX_HW_QUEUES=5 run both_q5 f16gpx3 '{"xstep": true, "gp_stream": true}'
GPU_MAX_HW_QUEUES=6 run both_q6 f16gpx3 '{"xstep": true, "gp_stream": true}'
