#!/bin/bash
# round 5, final build: GPU test-suite, the driver-form bench line, the other configurations, one rank's share of the sharded update
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
( time python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider ) > gpurun_out/r05/pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05/pytest_gpu.log; tail -4 gpurun_out/r05/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r05/bench_n1_detail.json > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err
echo "bench rc=$?"; tail -c 400 gpurun_out/r05/bench_n1.json
python scripts/bench_extra.py > gpurun_out/r05/bench_extra.jsonl 2>/dev/null
python scripts/bench_extra.py --shard-of 2,4,8 --precision f16gpx3 > gpurun_out/r05/shard_compute.jsonl 2>/dev/null
python scripts/bench_extra.py --shard-of 2,4,8 --precision bf16 >> gpurun_out/r05/shard_compute.jsonl 2>/dev/null
python bench.py --gpus 1 --steps 10 --warmup 3 --modes f16,f16gp32,bf16 --no-config5 --detail gpurun_out/r05/modes_parity_detail.json > gpurun_out/r05/modes.json 2>/dev/null
echo done
