#!/bin/bash
# Round 6, GPU call G: ppo_head at 64 registers (co-resides with the phased NT kernel's 64 free registers) against the same build with
# the 72-register kernel (libase_hip_q0.so); then the GPU tests that exercise it.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_agent.py -x -q -m gpu > $O/pytest_gpu_subset.txt 2>&1
tail -3 $O/pytest_gpu_subset.txt
REPS=3 timeout 2400 bash scripts/lab/ab_lib.sh libase_hip_q0.so libase_hip.so f16gpx3 > $O/ab_ppohead_f16gpx3.txt 2>&1
grep update $O/ab_ppohead_f16gpx3.txt
