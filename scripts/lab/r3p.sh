#!/bin/bash
O=gpurun_out/r3p; mkdir -p $O; rm -f $O/spikes.log
nproc >> $O/spikes.log; cat /proc/loadavg >> $O/spikes.log
for v in 1 0 1 0; do
  echo "== HSA_ENABLE_INTERRUPT=$v" >> $O/spikes.log
  HSA_ENABLE_INTERRUPT=$v timeout 300 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --verbose 2>&1 >/dev/null | grep "per-update" >> $O/spikes.log
  grep steal /proc/stat | head -1 >> $O/spikes.log; head -1 /proc/stat >> $O/spikes.log
done
cat $O/spikes.log
