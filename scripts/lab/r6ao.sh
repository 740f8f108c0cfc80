#!/bin/bash
# Round 6, GPU call AO: the GPU suite + smoke on the FINAL tree (last call of the round).
cd "$(dirname "$0")/../.."
O=gpurun_out/r6ao; mkdir -p $O
sha256sum ase_amd/csrc/libase_hip.so > $O/lib_sha256.txt; cat $O/lib_sha256.txt
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
