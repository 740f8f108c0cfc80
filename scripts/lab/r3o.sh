#!/bin/bash
O=gpurun_out/r3o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -k self_consistent > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 --breakdown --verbose > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python bench.py --precision f16 --steps 20 --warmup 5 --no-cpu-baseline --verbose > $O/bench_f16.json 2> $O/bench_f16.err
timeout 900 bash scripts/profile_round.sh; echo "prof rc=$?" >> $O/rc.txt
timeout 600 python scripts/bench_extra.py > $O/extra.jsonl 2> $O/extra.err; echo "extra rc=$?" >> $O/rc.txt
tail -3 $O/tests.log; cat $O/rc.txt; grep -o '"ms_per_step": [0-9.]*' $O/bench_bf16.json | head -1; grep -o '"ms_per_step": [0-9.]*' $O/bench_f16.json | head -1
