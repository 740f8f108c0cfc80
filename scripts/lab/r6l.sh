#!/bin/bash
# Round 6, GPU call L: hipGraph capture crash (calls J / K: the short prologue is the ingredient) - is it the hipMemsetAsync node on the
# forked stream?  (ASE_DEBUG_ZERO=torch: the gradient fill as a torch kernel instead)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6l; mkdir -p $O
B="python bench.py --gpus 1 --steps 3 --warmup 1 --hipgraph --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail ''"
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$? $(tail -1 $O/$n.json | cut -c1-200 | grep -o 'ms_per_step[^,]*')"; }
ASE_DEBUG_ZERO=torch run torchzero_bf16 $B --precision bf16
ASE_DEBUG_ZERO=torch run torchzero_gpx3 $B --precision f16gpx3
run program_bf16 python bench.py --gpus 1 --steps 3 --warmup 1 --precision bf16 --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail ''
