#!/bin/bash
L=scripts/lab/gemm_lab
export LAB_PROF=1
for wg in 256 128 64; do
  for shape in "16384 1024 1024" "16384 512 1024"; do
    echo -n "target_wg $wg: "; ASE_TN8_TARGET_WG=$wg timeout 60 $L tn $shape 20 0 || echo "rc=$?"
  done
done
