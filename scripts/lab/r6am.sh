#!/bin/bash
# Round 6, GPU call AM: stream priorities under mixed_precision: True (the dynamic loss scale makes the discriminator branch the
# step's critical path): default [critic high] against [discriminator high] / [none] / [both], three interleaved repetitions.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6am; mkdir -p $O; : > $O/sweep.txt
run() { n=$1; shift; ms=$(timeout 300 python scripts/bench_extra.py --only ase-mixed --updates 8 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_update'])"); echo "$n $ms" | tee -a $O/sweep.txt; }
for rep in 1 2 3; do
  run base
  run prio_disc_high --engine-opts '{"side_priority": [0, -1, 0]}'
  run prio_none --engine-opts '{"side_priority": [0, 0, 0]}'
  run prio_both_high --engine-opts '{"side_priority": [-1, -1, 0]}'
done
