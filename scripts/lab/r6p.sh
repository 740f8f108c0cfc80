#!/bin/bash
# Round 6, GPU call P: the GPU suite on the final tree (after call O's one failure: the new tail-count bound of the f32 schedule test
# was set at the edge of its own benign drift).
cd "$(dirname "$0")/../.."
O=gpurun_out/r6p; mkdir -p $O
sha256sum ase_amd/csrc/libase_hip.so > $O/lib_sha256.txt
timeout 1800 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
