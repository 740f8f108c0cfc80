#!/bin/bash
# one gpurun call of round 5: GPU tests + the driver-form bench line (+ whatever extra commands are passed)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
( time python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider ) > gpurun_out/r05/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05/pytest_gpu.log
tail -5 gpurun_out/r05/pytest_gpu.log
( time python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r05/bench_detail.json ) > gpurun_out/r05/bench.out 2> gpurun_out/r05/bench.err
echo "bench rc=$?"
tail -c 6000 gpurun_out/r05/bench.out
for c in "$@"; do echo "== $c"; bash -c "$c"; done
