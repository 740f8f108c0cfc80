#!/bin/bash
L=scripts/lab/gemm_lab
echo "--- NT schedule variants"
for sched in 0 1 2 3; do
  for shape in "16384 1024 1024" "8192 8192 8192"; do
    echo -n "sched $sched: "; ASE_NT_TILE=256 ASE_NT8_SCHED=$sched timeout 60 $L nt $shape 20 0 1 || echo "rc=$?"
  done
done
echo "--- TN phased vs 128x128"
for shape in "16384 1024 1024" "32768 1024 1024" "16384 512 1024" "32768 512 1024" "16384 1024 1408" "32768 1024 320" "32768 256 512" "4096 1024 1408" "12288 1024 1024" "8192 4096 4096"; do
  for v in 1 0; do
    echo -n "tn8=$v: "; ASE_TN8=$v timeout 60 $L tn $shape 20 1 || echo "rc=$?"
  done
done
echo "--- TN edge shapes"
for shape in "1024 128 128" "1088 200 136" "2048 1000 320" "4096 130 520"; do
  ASE_TN8=1 timeout 60 $L tn $shape 5 1 || echo "rc=$?"
done
