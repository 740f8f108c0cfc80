#!/bin/bash
O=gpurun_out/r3r; mkdir -p $O; rm -f $O/spikes.log
cat /sys/fs/cgroup/cpu.max >> $O/spikes.log; python -c "import torch; print('torch threads', torch.get_num_threads())" >> $O/spikes.log
for t in 256 1 256 1; do
  echo "== host threads $t" >> $O/spikes.log
  timeout 300 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --verbose --host-threads $t 2>&1 >/dev/null | grep "bench\]" | cut -c1-600 >> $O/spikes.log
done
cat $O/spikes.log
