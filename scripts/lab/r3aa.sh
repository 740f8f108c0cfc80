#!/bin/bash
O=gpurun_out/r3aa; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -q -k "identity_map or f32_path or f16gp or gather" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 200 python bench.py --precision f16gpx3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
tail -3 $O/tests.log; cat $O/rc.txt; grep -o '"ms_per_step": [0-9.]*' $O/bench.json | head -1
