#!/bin/bash
# round 5, the last GPU call: the whole GPU suite on the ABI-5 build (309 tests), then config 2 under the reference's mixed_precision flag
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05f
( time timeout 186 python -m pytest tests -x -q -m gpu -p no:cacheprovider ) > gpurun_out/r05f/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05f/pytest_gpu.log; tail -6 gpurun_out/r05f/pytest_gpu.log
( timeout 45 python scripts/bench_extra.py --only ase-mixed ) > gpurun_out/r05f/bench_mixed.jsonl 2> gpurun_out/r05f/bench_mixed.err
echo "mixed rc=$?"; cat gpurun_out/r05f/bench_mixed.jsonl
