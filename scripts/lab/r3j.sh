#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --breakdown > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 300 python bench.py --precision f16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err
tail -4 $O/tests.log; cat $O/rc.txt; grep -o '"ms_per_step": [0-9.]*' $O/bench_bf16.json $O/bench_f16.json; grep "^nt8" $O/bench_bf16.err | head -14
