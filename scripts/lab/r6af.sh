#!/bin/bash
# Round 6, GPU call AF: the driver-form bench line after (a) the gp_stream 'auto' rule reads THIS rank's rows and (b) a gp_f32 engine
# keeps taking four streams from the pool (later engines of the process keep their hardware-queue places): headline AND the later legs.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6af; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_n1_detail.json > $O/bench_n1.json 2> $O/bench_n1.err
python - <<PY
import json
d=json.loads(open('$O/bench_n1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], (d.get('throughput_mode') or {}).get('ms_per_step'), (d.get('strict_mode') or {}).get('ms_per_step'), (d.get('config5_16384_envs') or {}).get('ms_per_step'), d['roofline'].get('sustained_clock_mhz'))
PY
timeout 600 python scripts/bench_extra.py --shard-of 2,4,8 --precision f16gpx3 --updates 6 > $O/shard_compute.jsonl 2> $O/shard_compute.err
timeout 600 python scripts/bench_extra.py --shard-of 2,4,8 --precision bf16 --updates 6 >> $O/shard_compute.jsonl 2>> $O/shard_compute.err
grep -o '"ranks": [0-9]*, "ms_per_update": [0-9.]*, "us_per_step": [0-9.]*\|"gp_stream": [a-z]*\|"precision": "[a-z0-9]*"' $O/shard_compute.jsonl | paste - - - 
