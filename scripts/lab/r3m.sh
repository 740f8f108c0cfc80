#!/bin/bash
O=gpurun_out/r3m; mkdir -p $O
for g in on freeze on freeze; do
  echo "== gc $g" >> $O/spikes.log
  timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --verbose --gc $g 2>&1 >/dev/null | grep "bench\]" >> $O/spikes.log
done
rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -v "^$" | head -30 >> $O/spikes.log
cat $O/spikes.log
