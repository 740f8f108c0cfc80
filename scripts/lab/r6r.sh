#!/bin/bash
# Round 6, GPU call R: where one rank's share of the 8-way sharded step (2048 / 512 rows) spends its 620 us - kernel trace + one step's
# launches by queue and start time.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6r; mkdir -p $O
R=/tmp/prof_r6r; rm -rf $R
for P in f16gpx3 bf16; do
  rocprofv3 --kernel-trace --stats -d $R/$P -o t -- python scripts/bench_extra.py --shard-of 8 --precision $P --updates 4 > $O/shard8_$P.log 2>&1
  python scripts/rocpd_stats.py $R/$P/t_results.db 45 > $O/kernel_stats_shard8_$P.txt
  python scripts/rocpd_timeline.py $R/$P/t_results.db 120 > $O/timeline_shard8_$P.txt
  rm -rf $R/$P
done
tail -2 $O/shard8_f16gpx3.log | cut -c1-300
