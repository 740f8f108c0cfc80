#!/bin/bash
# Round 6, GPU call M: (1) the hipGraph tests + the kept hipGraph-replay bench line (graph_capture: 'hipgraph' now keeps the serial
# prologue), (2) the driver-form bench line once more, now that profiles/r06_pmc.json (measured on this very library) exists: the
# line quotes roofline.traffic.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6m; mkdir -p $O
sha256sum ase_amd/csrc/libase_hip.so > $O/lib_sha256.txt
timeout 900 python -m pytest tests -x -q -m gpu -k "hipgraph or graph" > $O/pytest_hipgraph.txt 2>&1; tail -2 $O/pytest_hipgraph.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --precision f16gpx3 --hipgraph --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --detail '' > $O/bench_hipgraph.json 2> $O/bench_hipgraph.err
tail -1 $O/bench_hipgraph.json | cut -c1-400
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --precision bf16 --hipgraph --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail '' > $O/bench_hipgraph_bf16.json 2> $O/bench_hipgraph_bf16.err
tail -1 $O/bench_hipgraph_bf16.json | cut -c1-400
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_n1_detail.json > $O/bench_n1.json 2> $O/bench_n1.err
tail -1 $O/bench_n1.json | cut -c1-1200
