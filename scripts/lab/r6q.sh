#!/bin/bash
# Round 6, GPU call Q: RCCL itself under this round's exchange (three collectives per step inside recorded launch programs, loss sums in
# the policy bucket, the merged statistics collective): a ONE-rank nccl group (bench.py --force-dist) - all this pool can give RCCL.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6q; mkdir -p $O
C="--gpus 1 --steps 5 --warmup 3 --force-dist --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode"
MASTER_ADDR=127.0.0.1 MASTER_PORT=29521 timeout 600 python bench.py $C --precision f16gpx3 > $O/force_dist_gpx3.json 2> $O/force_dist_gpx3.err; echo "gpx3 rc=$?"; tail -1 $O/force_dist_gpx3.json | cut -c1-400
MASTER_ADDR=127.0.0.1 MASTER_PORT=29522 timeout 600 python bench.py $C --precision bf16 --no-parity-mode > $O/force_dist_bf16.json 2> $O/force_dist_bf16.err; echo "bf16 rc=$?"; tail -1 $O/force_dist_bf16.json | cut -c1-300
MASTER_ADDR=127.0.0.1 MASTER_PORT=29523 timeout 600 python bench.py $C --precision bf16 --no-parity-mode --dp-mode horovod > $O/force_dist_horovod.json 2> $O/force_dist_horovod.err; echo "horovod rc=$?"; tail -1 $O/force_dist_horovod.json | cut -c1-300
grep -h "dist" $O/force_dist_gpx3.json | grep -o '"dist": {[^}]*}' | head -2
