#!/bin/bash
# A/B of the phased 256x256 NT kernel against the lock-step one (ASE_NT_VARIANT=3) + edge shapes
L=scripts/lab/gemm_lab
for shape in "16384 1024 1024" "32768 1024 1024" "12288 1024 1408" "32768 512 1024" "16384 1024 512" "4096 4096 4096" "8192 8192 8192"; do
  for v in 1 0; do
    echo -n "variant $v: "; ASE_NT_TILE=256 ASE_NT_PHASED=$v timeout 60 $L nt $shape 20 0 1 || echo "rc=$?"
  done
done
echo "--- edge shapes (forced 256 tile, phased kernel)"
for shape in "300 260 64" "1000 520 128" "16000 1000 192" "777 256 256" "4096 1408 1024"; do
  ASE_NT_TILE=256 timeout 60 $L nt $shape 5 1 1 || echo "rc=$?"
done
echo "--- with aux mask"
ASE_NT_TILE=256 timeout 60 $L nt 16384 1024 1024 20 1 0
ASE_NT_TILE=256 ASE_NT_VARIANT=3 timeout 60 $L nt 16384 1024 1024 20 1 0
