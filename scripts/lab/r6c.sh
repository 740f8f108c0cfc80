#!/bin/bash
# Round 6, GPU call C: whole GPU suite on the build with the 4-wave NT kernel + packed epilogue, then the same-box A/B against the
# round-5 library (ase_amd/csrc/libase_hip_r5.so = HEAD of round 5 built from `git archive`).
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6c
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6c/pytest_gpu.txt 2>&1
tail -3 gpurun_out/r6c/pytest_gpu.txt
REPS=2 timeout 1500 bash scripts/lab/ab_lib.sh libase_hip_r5.so libase_hip.so f16gpx3 > gpurun_out/r6c/ab_f16gpx3.txt 2>&1
REPS=1 timeout 900 bash scripts/lab/ab_lib.sh libase_hip_r5.so libase_hip.so bf16 > gpurun_out/r6c/ab_bf16.txt 2>&1
tail -4 gpurun_out/r6c/ab_f16gpx3.txt
