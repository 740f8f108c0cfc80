#!/bin/bash
L=scripts/lab/gemm_lab
for aux in 0 1 2 3; do
  for shape in "16384 1024 1024" "32768 1024 512"; do
    echo -n "aux=$aux: "; LAB_PROF=1 timeout 60 $L nt $shape 20 $aux 1 || echo "rc=$?"
  done
done
