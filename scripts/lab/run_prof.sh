#!/bin/bash
L=scripts/lab/gemm_lab
export ASE_NT_TILE=256 LAB_PROF=1
for shape in "16384 1024 1024" "32768 1024 1024" "16384 1024 64" "16384 1024 512" "8192 8192 1024"; do
  for aux in 0 1; do timeout 60 $L nt $shape 20 $aux 1 || echo "rc=$?"; done
done
