#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "tn or grouped or weight" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=/tmp/prof_h; rm -rf $R
SER="python bench.py --no-graph --no-multi-stream --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $R/s -o t -- $SER > $O/bench_serial.log 2>&1
python scripts/rocpd_stats.py $R/s/t_results.db 14 > $O/kernel_stats_serial.txt
rm -rf $R/s
PMC="python bench.py --no-graph --no-multi-stream --steps 1 --warmup 0 --no-cpu-baseline"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d $R/p$i -o t -- $PMC > /dev/null 2>&1
  echo "## $c" >> $O/pmc_summary.txt
  python scripts/pmc_summary.py $R/p$i/t_results.db | head -12 >> $O/pmc_summary.txt
  rm -rf $R/p$i
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --breakdown --verbose > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 300 python bench.py --precision f16 --steps 10 --warmup 3 --no-cpu-baseline --verbose > $O/bench_f16.json 2> $O/bench_f16.err
tail -2 $O/tests.log; head -8 $O/kernel_stats_serial.txt | cut -c1-130; grep -h tn8g $O/pmc_summary.txt; grep -o '"ms_per_step": [0-9.]*' $O/bench_bf16.json $O/bench_f16.json
