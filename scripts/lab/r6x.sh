#!/bin/bash
# Round 6, GPU call X: the GPU suite on the final tree (332 tests).
cd "$(dirname "$0")/../.."
O=gpurun_out/r6x; mkdir -p $O
sha256sum ase_amd/csrc/libase_hip.so > $O/lib_sha256.txt
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
