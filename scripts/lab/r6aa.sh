#!/bin/bash
# Round 6, GPU call AA: the schedule options of the update engine re-measured on the final kernels (their defaults were tuned in rounds
# 4-5, before the matrix kernels left 64 registers per SIMD lane to co-resident kernels): one box, base interleaved, f16gpx3.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6aa; mkdir -p $O; : > $O/sweep.txt
B="python bench.py --gpus 1 --steps 12 --warmup 4 --precision f16gpx3 --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail ''"
run() { n=$1; shift; ms=$(timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"); echo "$n $ms" | tee -a $O/sweep.txt; }
for rep in 1 2; do
  run base
  run tn_wg_side_32 --engine-opts '{"tn_wg_side": 32}'
  run tn_wg_side_128 --engine-opts '{"tn_wg_side": 128}'
  run tn_wg_side_0 --engine-opts '{"tn_wg_side": 0}'
  run base
  run tn_early --engine-opts '{"tn_early": true}'
  run style_early --engine-opts '{"style_early": true}'
  run prio_gp_high --engine-opts '{"side_priority": [0, 0, -1]}'
  run prio_disc_high --engine-opts '{"side_priority": [0, -1, 0]}'
  run base
  run one_side_stream --engine-opts '{"side_streams": 1}'
  run no_gp_stream --engine-opts '{"gp_stream": false}'
done
