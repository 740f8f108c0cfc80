#!/bin/bash
# Round 6, GPU call K: (1) hipGraph capture of the multi-stream benchmark step segfaults in capture_end (call J: every precision; ONE
# stream works) - which schedule option it takes; (2) tn_reduce at 65 registers (libase_hip_t65.so) against the shipped 64-register cap.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6k; mkdir -p $O
B="python bench.py --gpus 1 --steps 2 --warmup 1 --hipgraph --precision bf16 --no-cpu-baseline --no-config5 --throughput-mode none --no-strict-mode --no-parity-mode --detail ''"
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2> $O/$n.err; echo "$n rc=$? $(tail -1 $O/$n.json | cut -c1-200 | grep -o 'ms_per_step[^,]*')"; }
run tn_direct $B --engine-opts '{"tn_grouped": false}'
run long_prologue $B --engine-opts '{"short_prologue": false}'
run disc_late $B --engine-opts '{"disc_early": false}'
run one_side $B --engine-opts '{"side_streams": 1}'
run unfused_apply $B --engine-opts '{"fused_apply": false}'
run no_bits $B --engine-opts '{"relu_bits": false}'
ASE_DEBUG_NO_ACC_MARK=1 run no_acc_mark $B
run all_late $B --engine-opts '{"short_prologue": false, "disc_early": false, "side_streams": 1}'
HIP_LAUNCH_BLOCKING=0 AMD_SERIALIZE_KERNEL=0 run plain $B
REPS=3 timeout 1200 bash scripts/lab/ab_lib.sh libase_hip_t65.so libase_hip.so f16gpx3 > $O/ab_tnreduce_cap.txt 2>&1
grep update $O/ab_tnreduce_cap.txt
