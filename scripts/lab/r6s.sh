#!/bin/bash
# Round 6, GPU call S: is one rank's share of the sharded step bound by the host's replay loop?  (scripts/lab/host_vs_gpu.py)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6s; mkdir -p $O
timeout 900 python scripts/lab/host_vs_gpu.py f16gpx3,bf16 1,2,4,8 > $O/host_vs_gpu.jsonl 2> $O/host_vs_gpu.err
cat $O/host_vs_gpu.jsonl
