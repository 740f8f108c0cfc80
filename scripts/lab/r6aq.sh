#!/bin/bash
# Round 6, GPU call AQ: the contract's N > 1 launch line once more on the FINAL tree (two ranks sharing the one GPU, gloo), sharded
# headline + Horovod, and the one-rank RCCL group.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6aq; mkdir -p $O
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
C="--gpus 2 --steps 3 --warmup 2 --dist-backend gloo --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode"
timeout 900 $L --master-port 29531 bench.py $C > $O/dist2_auto.json 2> $O/dist2_auto.err; echo "auto rc=$?"; tail -1 $O/dist2_auto.json | cut -c1-200
timeout 600 $L --master-port 29532 bench.py $C --precision bf16 --no-parity-mode --dp-mode horovod > $O/dist2_horovod.json 2> $O/dist2_horovod.err; echo "horovod rc=$?"; tail -1 $O/dist2_horovod.json | cut -c1-200
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 600 python bench.py --gpus 1 --steps 5 --warmup 3 --force-dist --precision f16gpx3 --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode > $O/force_dist.json 2> $O/force_dist.err; echo "rccl-1 rc=$?"; tail -1 $O/force_dist.json | cut -c1-200
