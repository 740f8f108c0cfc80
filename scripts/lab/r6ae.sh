#!/bin/bash
# Round 6, GPU call AE: same-box A/B of engine_opts gp_stream on one rank's share of the sharded update (R = 2, 4, 8; f16gpx3), three
# interleaved repetitions - the boxes of calls AC / AD differ too much at these launch-latency-bound sizes to compare across calls;
# then the driver-form bench line once more.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6ae; mkdir -p $O; : > $O/shard_ab.jsonl
for rep in 1 2 3; do
  for v in true false; do
    timeout 300 python scripts/bench_extra.py --shard-of 2,4,8 --precision f16gpx3 --updates 6 --engine-opts "{\"gp_stream\": $v}" 2>/dev/null >> $O/shard_ab.jsonl
  done
done
grep -o '"ranks": [0-9]*, "ms_per_update": [0-9.]*, "us_per_step": [0-9.]*\|"gp_stream": [a-z]*' $O/shard_ab.jsonl | paste - - | sort | uniq -c | head -0
python - <<PY
import json
rows=[json.loads(l) for l in open('$O/shard_ab.jsonl')]
for R in (2,4,8):
    for v in (True, False):
        xs=[r['us_per_step'] for r in rows if r['ranks']==R and r['gp_stream']==v]
        print('R', R, 'gp_stream', v, xs)
PY
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_n1_detail.json > $O/bench_n1.json 2> $O/bench_n1.err
tail -1 $O/bench_n1.json | cut -c1-400
