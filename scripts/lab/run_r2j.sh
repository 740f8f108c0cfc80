#!/bin/bash
# Round 2, sweep J: one wave per SIMD - 256 x 256 tile as FOUR waves of 128 x 128 (v30), 256 x 128 tile with a 3-stage ring (v32),
# both compiler-scheduled lock-step kernels with the row-per-lane epilogue, against the phased kernel (v0)
L=scripts/lab/gemm_lab
for s in "8192 8192 8192" "16384 1024 1024" "32768 1024 1024" "16384 1024 512" "32768 1024 320" "12288 1024 1408" "32768 512 64"; do
  for v in 0 30 32; do
    echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 20 2 0 | tail -1 || echo "rc=$?"
  done
done
