#!/bin/bash
O=gpurun_out/r3v; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
# the driver's launch line with ONE rank over RCCL (collectives inside the launch program)
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 5 --warmup 2 --force-dist --no-cpu-baseline > $O/fd.json 2> $O/fd.err; echo "force-dist rc=$?" > $O/rc.txt
# the N > 1 flow: 2 ranks sharing the one GPU (gloo), horovod (default) and shard
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --dist-backend gloo > $O/n2.json 2> $O/n2.err; echo "n2 horovod rc=$?" >> $O/rc.txt
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 2 --steps 3 --warmup 1 --dist-backend gloo --dp-mode shard > $O/n2s.json 2> $O/n2s.err; echo "n2 shard rc=$?" >> $O/rc.txt
cat $O/rc.txt; for f in fd n2 n2s; do grep -o '"value": [0-9.]*, "unit": "samples/s", "n_gpus": [0-9]*' $O/$f.json | head -1; grep -o '"scaling": "[a-z]*"' $O/$f.json | head -1; tail -2 $O/$f.err | cut -c1-200; done
