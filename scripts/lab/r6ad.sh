#!/bin/bash
# Round 6, GPU call AD: engine_opts gp_stream 'auto' (own stream for the penalty's value path below 8192 minibatch rows and under the
# dynamic loss scale, else not): the side measurements call AC lost with a plain 'off' must be back - and the headline unchanged.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6ad; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_scaler.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
timeout 600 python scripts/bench_extra.py --only ase-dyn-gpx3,ase-mixed --updates 8 > $O/bench_dyn.jsonl 2> $O/bench_dyn.err; cut -c1-200 $O/bench_dyn.jsonl
timeout 600 python scripts/bench_extra.py --shard-of 2,4,8 --precision f16gpx3 --updates 6 > $O/shard_compute.jsonl 2> $O/shard_compute.err
timeout 600 python scripts/bench_extra.py --shard-of 2,4,8 --precision bf16 --updates 6 >> $O/shard_compute.jsonl 2>> $O/shard_compute.err
grep -o '"ranks": [0-9]*, "ms_per_update": [0-9.]*, "us_per_step": [0-9.]*' $O/shard_compute.jsonl
timeout 300 python bench.py --gpus 1 --steps 12 --warmup 4 --precision f16gpx3 --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail '' 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline', d['ms_per_step'])"
