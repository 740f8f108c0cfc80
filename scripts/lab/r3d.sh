cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
L=scripts/lab/gemm_lab
( for ws in 0 1; do echo "== tng ws=$ws"; LAB_TNG_WS=$ws LAB_PROF=1 timeout 120 $L tng 10 | grep -v "): maxerr"; done ) > $O/tng.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --breakdown > $O/bench.json 2> $O/bench.err
timeout 600 python scripts/bench_extra.py --shard-of 2,4,8 > $O/shard.jsonl 2> $O/shard.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=/tmp/prof_r3d; rm -rf $R
rocprofv3 --kernel-trace --stats -d $R/s -o t -- python bench.py --no-graph --no-multi-stream --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_serial.log 2>&1
python scripts/rocpd_stats.py $R/s/t_results.db 45 > $O/kernel_stats_serial.txt
rm -rf $R/s
rocprofv3 --kernel-trace --stats -d $R/g -o t -- python bench.py --no-cpu-baseline > $O/bench_replay.log 2>&1
python scripts/rocpd_stats.py $R/g/t_results.db 45 > $O/kernel_stats_replay.txt
python scripts/rocpd_timeline.py $R/g/t_results.db 70 > $O/timeline_replay.txt
rm -rf $R
tail -5 $O/pytest.log
