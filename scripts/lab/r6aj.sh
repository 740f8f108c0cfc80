#!/bin/bash
# Round 6, GPU call AJ: the penalty's value path behind the loss rows' forward and heads (engine_opts gp_value_late) against in front
# of them (default), f16gpx3, four interleaved repetitions; correctness of the late order through the gp_f32 engine tests.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6aj; mkdir -p $O; : > $O/sweep.txt
run() { n=$1; shift; ms=$(timeout 300 python bench.py --gpus 1 --steps 12 --warmup 4 --precision f16gpx3 --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --detail '' "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], (d.get('parity') or {}).get('fresh', {}).get('max_loss_term_rel'), (d.get('parity') or {}).get('stress', {}).get('max_loss_term_rel'))"); echo "$n $ms" | tee -a $O/sweep.txt; }
for rep in 1 2 3 4; do
  run base
  run gp_value_late --engine-opts '{"gp_value_late": true}'
done
