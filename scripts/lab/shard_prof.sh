#!/bin/bash
# rocprofv3 kernel statistics of ONE rank's share of the row-sharded update (scripts/bench_extra.py --shard-of R) for R = 4 and 2:
# where the time of a small-minibatch step goes (12-18 us launches).  Through gpurun: bash scripts/lab/shard_prof.sh -> gpurun_out/shard<R>_kstats.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for R in ${RANKS:-4 2 8}; do
rm -rf /tmp/ps$R
rocprofv3 --kernel-trace --stats -d /tmp/ps$R -o t -- python scripts/bench_extra.py --shard-of $R --updates 3 --precision ${PREC:-f16gpx3} > gpurun_out/shard${R}_prof.log 2>&1
python scripts/rocpd_stats.py /tmp/ps$R/t_results.db 25 > gpurun_out/shard${R}_kstats.txt
done
