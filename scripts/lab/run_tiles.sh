#!/bin/bash
# tile choice per in-situ shape: forced 128 x 128 vs forced phased 256 x 256 (aux = bit mask)
L=scripts/lab/gemm_lab
for shape in "12288 1024 1408" "12288 1024 1024" "12288 512 1024" "16384 512 1024" "16384 1024 320" "32768 1024 320" \
             "4096 1024 1408" "4096 1408 1024" "4096 512 1024" "4096 1024 1024" "32768 256 512" "32768 512 256" \
             "32768 512 64" "16384 512 64" "32768 256 64" "12288 512 128" "12288 128 512"; do
  for t in 128 256; do
    echo -n "tile $t: "; ASE_NT_TILE=$t timeout 60 $L nt $shape 20 2 0 || echo "rc=$?"
  done
done
