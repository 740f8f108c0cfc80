#!/bin/bash
# Round 6, GPU call AH: second pass over the schedule options on the final tree (gp_stream 'auto' = off in the headline regime), f16gpx3
# and bf16, base interleaved.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6ah; mkdir -p $O; : > $O/sweep.txt
run() { n=$1; p=$2; shift 2; ms=$(timeout 300 python bench.py --gpus 1 --steps 12 --warmup 4 --precision $p --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail '' "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"); echo "$p $n $ms" | tee -a $O/sweep.txt; }
for rep in 1 2; do
  for p in f16gpx3 bf16; do
    run base $p
    run prio_critic_high $p --engine-opts '{"side_priority": [-1, 0, 0]}'
    run prio_none $p --engine-opts '{"side_priority": [0, 0, 0]}'
    run prio_both_high $p --engine-opts '{"side_priority": [-1, -1, 0]}'
    run base $p
    run disc_after_style $p --engine-opts '{"disc_after_style": true}'
    run no_xstep $p --engine-opts '{"xstep": false}'
    run tn_wg_side_96 $p --engine-opts '{"tn_wg_side": 96}'
    run style_side_wg64 $p --engine-opts '{"style_side": 1, "style_wg": 64}'
  done
done
