#!/bin/bash
# Round 6, GPU call AN: after the priority default under the dynamic loss scale: the scaler / agent GPU tests, mixed_precision and the
# headline mode under the dynamic scale once more.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6an; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_scaler.py tests/test_gpu_agent.py -x -q -m gpu > $O/pytest_subset.txt 2>&1; tail -2 $O/pytest_subset.txt
timeout 600 python scripts/bench_extra.py --only ase-mixed,ase-dyn-gpx3 --updates 8 > $O/bench_dyn.jsonl 2> $O/bench_dyn.err; cut -c1-200 $O/bench_dyn.jsonl
