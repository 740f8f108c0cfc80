#!/bin/bash
O=gpurun_out/r3t; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "f32_path or f16gp" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 300 python bench.py --precision f16gp32 --steps 10 --warmup 3 --no-cpu-baseline --verbose > $O/bench_gp32.json 2> $O/bench_gp32.err; echo "bench rc=$?" >> $O/rc.txt
tail -15 $O/tests.log; cat $O/rc.txt; grep -o '"ms_per_step": [0-9.]*' $O/bench_gp32.json | head -1; grep "per-update" $O/bench_gp32.err | cut -c1-200
