cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 --breakdown > $O/bench.json 2> $O/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --precision f16 --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=/tmp/prof_r3f; rm -rf $R
rocprofv3 --kernel-trace --stats -d $R/s -o t -- python bench.py --no-graph --no-multi-stream --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_serial.log 2>&1
python scripts/rocpd_stats.py $R/s/t_results.db 45 > $O/kernel_stats_serial.txt
rm -rf $R
tail -5 $O/pytest.log
