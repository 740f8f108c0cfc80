#!/bin/bash
# Round 6, GPU call F: the phased NT / TN kernels with `buffer_load ... lds` DMA (217 / 205 registers instead of 238: 64 / 96 free per
# SIMD for co-resident small kernels).  GPU tests of the matrix operators + agents, then the same-box A/B against p0 (the same build
# with the 64-bit-address DMA), four streams and one.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6f; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
REPS=3 timeout 2400 bash scripts/lab/ab_lib.sh libase_hip_p0.so libase_hip.so f16gpx3 > $O/ab_bufdma_f16gpx3.txt 2>&1
grep update $O/ab_bufdma_f16gpx3.txt
REPS=2 timeout 1500 bash scripts/lab/ab_lib.sh libase_hip_p0.so libase_hip.so bf16 > $O/ab_bufdma_bf16.txt 2>&1
grep update $O/ab_bufdma_bf16.txt
