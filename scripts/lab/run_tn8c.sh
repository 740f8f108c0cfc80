#!/bin/bash
L=scripts/lab/gemm_lab
export LAB_PROF=1 LAB_VERBOSE=1
for xl in 1 0; do
  for shape in "16384 1024 1024" "16384 512 1024" "32768 1024 1024"; do
    echo -n "xcd_local $xl: "; ASE_XCD_LOCAL=$xl timeout 60 $L tn $shape 20 1 || echo "rc=$?"
  done
done
