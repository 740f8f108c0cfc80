#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 --breakdown > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err
timeout 900 bash scripts/profile_round.sh; echo "prof rc=$?" >> $O/rc.txt
tail -3 $O/tests.log; cat $O/rc.txt; head -c 600 $O/bench_bf16.json
