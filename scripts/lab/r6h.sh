#!/bin/bash
# Round 6, GPU call H: the ABI-7 build (overflow detection inside the producers: scale records; dynamic-scale steps keep the
# cross-step schedule and the fused optimizer launch): (1) the GPU suite, (2) config 2 under mixed_precision: True (round 5: 82.5 ms,
# call D of this round: 86.0 ms with per-step GradScaler semantics but 0.82 GB of check reads per step and the end-of-step optimizer
# form), (3) the static f16 mode on the same box for the distance between the two, (4) the headline mode (records are NULL there:
# nothing may have moved).
cd "$(dirname "$0")/../.."
O=gpurun_out/r6h; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 600 python scripts/bench_extra.py --only ase-mixed --updates 8 > $O/bench_mixed.jsonl 2> $O/bench_mixed.err
tail -2 $O/bench_mixed.jsonl | cut -c1-700
for prec in f16 f16gpx3; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --precision $prec --no-cpu-baseline --no-config5 --throughput-mode none --detail '' > $O/bench_$prec.json 2> $O/bench_$prec.err
  python - <<PY
import json
d=json.loads(open('$O/bench_$prec.json').read().strip().splitlines()[-1])
print('$prec', d['ms_per_step'], 'ms/update', d['value'], d['unit'])
PY
done
