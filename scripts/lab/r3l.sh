#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or nt or tn" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 --breakdown --verbose > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --verbose > $O/bench_f16.json 2> $O/bench_f16.err
tail -2 $O/tests.log; cat $O/rc.txt; grep -o '"ms_per_step": [0-9.]*' $O/bench_bf16.json $O/bench_f16.json | head -4; grep -o '"sustained_clock_mhz.*"all_gemm"' $O/bench_bf16.json | cut -c1-400; grep "update" $O/bench_f16.err | head -30
