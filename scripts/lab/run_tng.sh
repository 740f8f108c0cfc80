#!/bin/bash
L=scripts/lab/gemm_lab
LAB_PROF=1 timeout 120 $L tng 10 || echo "rc=$?"
echo "--- single TN with bias, phased vs 128"
for shape in "32768 1024 1024" "16384 1024 1024"; do
  for v in 1 0; do echo -n "tn8=$v: "; ASE_TN8=$v timeout 60 $L tn $shape 20 1 || echo "rc=$?"; done
done
