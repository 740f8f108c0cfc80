#!/bin/bash
# round 5, last GPU call: the dynamic loss scale on hardware (tests/test_gpu_scaler.py), smoke, a short default-path bench
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r05s
( time timeout 170 python -m pytest tests/test_gpu_scaler.py -q -m gpu -p no:cacheprovider --maxfail=20 ) > gpurun_out/r05s/pytest_scaler.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r05s/pytest_scaler.log; tail -5 gpurun_out/r05s/pytest_scaler.log
( timeout 60 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/r05s/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/r05s/smoke.log
( timeout 90 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 --throughput-mode none --precision f16gpx3 --detail gpurun_out/r05s/bench_detail.json ) > gpurun_out/r05s/bench_short.json 2> gpurun_out/r05s/bench_short.err
echo "bench rc=$?"; tail -c 300 gpurun_out/r05s/bench_short.json
