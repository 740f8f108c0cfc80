#!/bin/bash
# Round 6, GPU call E: (1) GPU tests touched since call D (loss-head torch ops, the merged statistics collective with two ranks on one
# GPU, scaler), (2) what the step's small kernels gain from co-residing with the matrix kernels: p0 = 8-wave NT kernel (480 of 512
# registers per SIMD), p3 = the same kernels made to own the whole register file (nothing co-resides), p1 = 4-wave kernel in the step;
# (3) the same three WITHOUT concurrency (one stream).
cd "$(dirname "$0")/../.."
O=gpurun_out/r6e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_torch_ops.py tests/test_gpu_agent.py tests/test_gpu_scaler.py tests/test_gpu_boundary.py -x -q -m gpu > $O/pytest_gpu_subset.txt 2>&1
tail -3 $O/pytest_gpu_subset.txt
REPS=2 timeout 2400 bash scripts/lab/ab_lib.sh libase_hip_p0.so libase_hip_p3.so libase_hip_p1.so f16gpx3 > $O/ab_coresidency_f16gpx3.txt 2>&1
grep update $O/ab_coresidency_f16gpx3.txt
AB_EXTRA='--no-multi-stream' REPS=1 timeout 1800 bash scripts/lab/ab_lib.sh libase_hip_p0.so libase_hip_p3.so libase_hip_p1.so f16gpx3 > $O/ab_onestream_f16gpx3.txt 2>&1
grep update $O/ab_onestream_f16gpx3.txt
