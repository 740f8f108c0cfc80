#!/bin/bash
L=scripts/lab/gemm_lab
for wg in 512 256 128 64; do
  for shape in "32768 64 512" "32768 64 256" "16384 64 512" "12288 64 512" "32768 512 64" "32768 256 512"; do
    echo -n "target_wg $wg: "; ASE_TN8=0 ASE_TN_TARGET_WG=$wg timeout 60 $L tn $shape 20 1 || echo "rc=$?"
  done
done
