#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p2; mkdir -p $O; R=/tmp/prof_x; rm -rf $R
rocprofv3 --kernel-trace --stats -d $R/s -o t -- python bench.py --precision f16gpx3 --no-graph --no-multi-stream --steps 2 --warmup 1 --no-cpu-baseline > $O/bench.log 2>&1
python scripts/rocpd_stats.py $R/s/t_results.db 24 > $O/kernel_stats_f16gpx3_serial.txt
head -14 $O/kernel_stats_f16gpx3_serial.txt | cut -c1-150
