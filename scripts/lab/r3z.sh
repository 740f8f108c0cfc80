#!/bin/bash
O=gpurun_out/r3z; mkdir -p $O; rm -f $O/log.txt
timeout 600 python -m pytest tests/test_gpu_agent.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/log.txt
timeout 300 python bench.py --steps 60 --warmup 3 --no-cpu-baseline --verbose 2>&1 >/dev/null | grep "bench\]" | cut -c1-500 >> $O/log.txt
tail -2 $O/tests.log; cat $O/log.txt
