#!/bin/bash
# The phased NT kernel beside the vendor library on the shapes that carry the update (same random operands, HIP events,
# back-to-back launches; rocBLAS: plain bf16 GEMM, ours: fused bias + ReLU + bit-mask epilogue).
#   make -C scripts/lab gemm_lab blas_ref && bash scripts/lab/vs_rocblas.sh > gpurun_out/vs_rocblas.txt
cd "$(dirname "$0")"
for s in "16384 1024 1024" "32768 1024 1024" "32768 1024 320" "16384 1024 512" "32768 512 1024" "12288 1024 1408" "131072 1024 1024"; do
  echo "== M N K = $s"
  ./gemm_lab nt $s 50 0 1 2>&1 | grep -E "us|TF" | head -3
  ./blas_ref $s 50 2>&1 | tail -1
done
for s in "16384 1024 1024" "32768 1024 1024"; do
  echo "== weight-gradient shape, M N K = $s"
  ./gemm_lab tn $s 30 2>&1 | grep -E "us|TF" | head -2
  ./blas_ref $s 30 tn 2>&1 | tail -1
done
