#!/bin/bash
# Round 6, GPU call T: host lanes in the launch-program replay (csrc/prog.cpp) - the GPU suite (every agent test replays programs over
# four streams; the 2-rank test replays host callbacks), then host vs GPU time of one rank's share with lanes on / off, and the
# benchmark update with lanes on / off.  Everything under timeouts: a lane that waits for ever must not take the box.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "launch_program" > $O/pytest_prog.txt 2>&1; tail -2 $O/pytest_prog.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 600 python scripts/lab/host_vs_gpu.py f16gpx3,bf16 1,2,4,8 > $O/host_vs_gpu_lanes.jsonl 2> $O/host_vs_gpu_lanes.err
ASE_PROG_LANES=0 timeout 600 python scripts/lab/host_vs_gpu.py f16gpx3,bf16 1,2,4,8 > $O/host_vs_gpu_serial.jsonl 2> $O/host_vs_gpu_serial.err
echo "## lanes"; cut -c1-330 $O/host_vs_gpu_lanes.jsonl; echo "## one thread"; cut -c1-330 $O/host_vs_gpu_serial.jsonl
for rep in 1 2; do
  for v in 1 0; do
    ASE_PROG_LANES=$v timeout 600 python bench.py --gpus 1 --steps 10 --warmup 3 --precision f16gpx3 --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail '' 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes=$v', d['ms_per_step'])"
  done
done
