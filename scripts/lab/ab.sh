#!/bin/bash
# Same-box A/B of engine schedule options (UpdateEngine.engine_opts) on the benchmark workload: every variant twice, interleaved.
#   bash scripts/lab/ab.sh <precision> '<engine_opts json>' ['<engine_opts json>' ...]  [-- extra bench.py flags]
# e.g. (through gpurun):  bash scripts/lab/ab.sh bf16 '{}' '{"prefetch": false}' '{"xstep": false, "prefetch": false}'
# Output: one line per run "<opts> :: ms_per_update samples/s".  The round's results are kept in profiles/r04_schedule_ab.txt.
cd "$(dirname "$0")/../.."
P=$1; shift
OPTS=(); EXTRA=()
while [ $# -gt 0 ]; do
  if [ "$1" == "--" ]; then shift; EXTRA=("$@"); break; fi
  OPTS+=("$1"); shift
done
for rep in 1 2; do
  for o in "${OPTS[@]}"; do
    out=$(python bench.py --gpus 1 --steps 20 --warmup 3 --precision "$P" --no-cpu-baseline --no-config5 --throughput-mode none --detail "" \
          --engine-opts "$o" "${EXTRA[@]}" 2>/dev/null | tail -1)
    echo "$P $o ${EXTRA[*]} :: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
