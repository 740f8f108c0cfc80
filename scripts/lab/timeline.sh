#!/bin/bash
# rocprofv3 kernel trace of the default (replayed, multi-stream) benchmark: one optimisation step's launches by queue and time.
# usage (through gpurun): bash scripts/lab/r4_timeline.sh <precision> [engine-opts json]  -> gpurun_out/timeline_<precision>.txt
P=${1:-bf16}; EO=${2:-{\}}
OUT=gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=/tmp/prof_tl_$P; rm -rf $R
rocprofv3 --kernel-trace --stats -d $R -o t -- python bench.py --precision $P --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --throughput-mode none --detail "" --engine-opts "$EO" > $OUT/timeline_${P}_bench.log 2>&1
python scripts/rocpd_timeline.py $R/t_results.db 60 > $OUT/timeline_$P.txt
python scripts/rocpd_stats.py $R/t_results.db 30 > $OUT/kstats_replay_$P.txt
rm -rf $R
