#!/bin/bash
# Round 6, GPU call AB: confirmation of call AA's one mover - the penalty's value path WITHOUT its own stream (engine_opts gp_stream:
# false: on the discriminator's stream) - base interleaved four times, and its combinations.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6ab; mkdir -p $O; : > $O/sweep.txt
B="python bench.py --gpus 1 --steps 12 --warmup 4 --precision f16gpx3 --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail ''"
run() { n=$1; shift; ms=$(timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"); echo "$n $ms" | tee -a $O/sweep.txt; }
for rep in 1 2 3 4; do
  run base
  run no_gp_stream --engine-opts '{"gp_stream": false}'
done
for rep in 1 2; do
  run no_gp_stream+prio_disc_high --engine-opts '{"gp_stream": false, "side_priority": [0, -1, 0]}'
  run no_gp_stream+one_side --engine-opts '{"gp_stream": false, "side_streams": 1}'
  run no_gp_stream+no_prefetch --engine-opts '{"gp_stream": false, "prefetch": false}'
  run no_gp_stream+tn_wg_side_128 --engine-opts '{"gp_stream": false, "tn_wg_side": 128}'
  run no_gp_stream --engine-opts '{"gp_stream": false}'
done
