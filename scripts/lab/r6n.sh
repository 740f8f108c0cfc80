#!/bin/bash
# Round 6, GPU call N: the hipGraph-replay bench line of the headline mode (graph_capture: 'hipgraph' keeps the serial prologue and
# the penalty's value path on the discriminator's stream: forks from forked streams crash torch's capture_end at this size).
cd "$(dirname "$0")/../.."
O=gpurun_out/r6n; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "hipgraph or graph" > $O/pytest_hipgraph.txt 2>&1; tail -2 $O/pytest_hipgraph.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --precision f16gpx3 --hipgraph --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --detail '' > $O/bench_hipgraph.json 2> $O/bench_hipgraph.err
echo rc=$?; tail -1 $O/bench_hipgraph.json | cut -c1-400; tail -3 $O/bench_hipgraph.err
