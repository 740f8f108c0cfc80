#!/bin/bash
# Round 6, GPU call J: which ingredient makes the hipGraph capture of the benchmark step crash in capture_end (call I: segmentation fault
# with --precision f16gpx3 --hipgraph).  Every variant under its own timeout; a crash does not end the script.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6j; mkdir -p $O
B="python bench.py --gpus 1 --steps 3 --warmup 1 --hipgraph --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail ''"
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$? $(tail -1 $O/$n.json | cut -c1-160)"; grep -h "Error\|error\|Segmentation\|hip" $O/$n.err | tail -3; }
run bf16 $B --precision bf16
run f16 $B --precision f16
run gpx3_nogpstream $B --precision f16gpx3 --engine-opts '{"gp_stream": false}'
run gpx3_onestream $B --precision f16gpx3 --no-multi-stream
run gp32 $B --precision f16gp32
run gpx3 $B --precision f16gpx3
