#!/bin/bash
# Round 6, GPU call AL: the penalty's value path (4-byte storage, 4096 rows) on 128 x 128 tiles - one workgroup per CU, like the phased
# kernels it runs beside (libase_hip_v128.so) - against the shipped 64 x 128 tiles (two per CU): in-step A/B, f16gpx3, three repetitions.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6al; mkdir -p $O
REPS=3 timeout 1500 bash scripts/lab/ab_lib.sh libase_hip.so libase_hip_v128.so f16gpx3 > $O/ab_f16gpx3.txt 2>&1; grep update $O/ab_f16gpx3.txt
