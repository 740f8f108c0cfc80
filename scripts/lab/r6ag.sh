#!/bin/bash
# Round 6, GPU call AG: the GPU suite + smoke on the final tree (gp_stream 'auto'), then: does the engine's PLACE on the hardware queues
# matter for the headline (engine_opts stream_offset: n throw-away streams per priority in front of the branch streams)?
cd "$(dirname "$0")/../.."
O=gpurun_out/r6ag; mkdir -p $O
sha256sum ase_amd/csrc/libase_hip.so > $O/lib_sha256.txt
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
: > $O/offset.txt
B="python bench.py --gpus 1 --steps 12 --warmup 4 --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail ''"
run() { n=$1; shift; ms=$(timeout 300 $B "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"); echo "$n $ms" | tee -a $O/offset.txt; }
for rep in 1 2; do
  for prec in f16gpx3 bf16; do
    for k in 0 1 2 3; do run "$prec offset=$k" --precision $prec --engine-opts "{\"stream_offset\": $k}"; done
  done
done
