#!/bin/bash
# Round 2, sweep G: small-M launches (data-parallel shards: 2048 rows per rank at 8 GPUs; gradient-penalty chain: 4096 rows)
L=scripts/lab/gemm_lab
for s in "2048 1024 1024" "2048 1024 320" "2048 512 1024" "4096 1024 1024" "4096 1024 1408" "4096 512 1024" "1536 1024 1408" "12288 512 1024" "16384 512 1024"; do
  for v in 0 20 21 22 23; do
    echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 20 2 0 | tail -1 || echo "rc=$?"
  done
done
