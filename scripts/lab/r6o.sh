#!/bin/bash
# Round 6, GPU call O: the contract's N > 1 launch line end to end with TWO ranks sharing the one GPU of the box (gloo; RCCL refuses two
# ranks on one device): the sharded update with this round's three collectives per step inside recorded launch programs, headline mode and
# bf16, and the Horovod mode; then the whole GPU suite once more (the engine's Python changed since call I: hipGraph mode).
cd "$(dirname "$0")/../.."
O=gpurun_out/r6o; mkdir -p $O
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
C="--gpus 2 --steps 3 --warmup 2 --dist-backend gloo --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode"
timeout 900 $L --master-port 29511 bench.py $C > $O/dist2_auto.json 2> $O/dist2_auto.err; echo "auto rc=$?"; tail -1 $O/dist2_auto.json | cut -c1-700
timeout 600 $L --master-port 29512 bench.py $C --precision bf16 --no-parity-mode > $O/dist2_bf16.json 2> $O/dist2_bf16.err; echo "bf16 rc=$?"; tail -1 $O/dist2_bf16.json | cut -c1-300
timeout 600 $L --master-port 29513 bench.py $C --precision bf16 --no-parity-mode --dp-mode horovod > $O/dist2_horovod.json 2> $O/dist2_horovod.err; echo "horovod rc=$?"; tail -1 $O/dist2_horovod.json | cut -c1-300
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
