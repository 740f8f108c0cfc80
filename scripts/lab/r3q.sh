#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O; rm -f $O/spikes.log
timeout 900 python -m pytest tests/test_gpu_agent.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
for v in 1 2; do
  timeout 300 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --verbose 2>&1 >/dev/null | grep "bench\]" | cut -c1-700 >> $O/spikes.log
done
timeout 600 python bench.py --steps 20 --warmup 5 --verbose > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench rc=$?" >> $O/rc.txt
tail -3 $O/tests.log; cat $O/rc.txt; cat $O/spikes.log; grep "bench\]" $O/bench_bf16.err | cut -c1-300; grep -o '"ms_per_step": [0-9.]*' $O/bench_bf16.json | head -1
