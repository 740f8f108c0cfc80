#!/bin/bash
# Round 6, GPU call B: the packed-math row-per-lane epilogue (nt8 and nt4v) - correctness incl. masks, time per shape; then the GPU
# tests of the NT operator on the product build.
cd "$(dirname "$0")"
OUT=../../gpurun_out/r6b; mkdir -p $OUT
T="timeout 120"
{
echo "#### 16384 x 1024 x 1024: product nt8 | nt4v (new epilogue), LAB_PROF"
LAB_PROF=1 $T ./gemm_lab nt 16384 1024 1024 50 0 1 2>&1 | grep -E "profile|TF|mismatch"
ASE_NT4V=2 LAB_PROF=1 $T ./gemm_lab nt 16384 1024 1024 50 0 1 2>&1 | grep -E "profile|TF|mismatch"
echo "#### masks / relu off / ragged: nt8 then nt4v"
for v in -1 2; do
  for a in "16384 1024 1024 30 2 1" "16384 1024 1024 30 3 1" "16384 1024 1024 30 0 0" "16300 1024 1024 30 3 1" "16300 1024 1024 30 2 0" "12288 1024 1408 30 3 1" "32768 1024 320 30 3 1" "16384 1024 64 20 3 1"; do
    if [ $v -ge 0 ]; then ASE_NT4V=$v $T ./gemm_lab nt $a 2>&1 | grep -E "TF|mismatch" | head -3; else $T ./gemm_lab nt $a 2>&1 | grep -E "TF|mismatch" | head -3; fi
  done
done
echo "#### smaller tiles with the row epilogue (128x128 / 64x128 / 64x64): forward mask + consumer"
for a in "16384 512 1024 30 3 1" "16384 512 1024 30 2 1" "4096 1024 1024 30 3 1" "4096 1024 1024 30 2 0" "4096 512 1024 30 3 1" "2048 1024 1024 30 2 1" "32768 512 256 30 3 1" "32768 256 512 30 3 1"; do
  $T ./gemm_lab nt $a 2>&1 | grep -E "TF|mismatch" | head -3
done
echo "#### the seven carrying shapes: product nt8 | nt4v | rocBLAS"
for s in "16384 1024 1024" "32768 1024 1024" "32768 1024 320" "16384 1024 512" "32768 512 1024" "12288 1024 1408" "131072 1024 1024"; do
  echo "== $s"; $T ./gemm_lab nt $s 50 3 1 2>&1 | grep TF; ASE_NT4V=2 $T ./gemm_lab nt $s 50 3 1 2>&1 | grep TF; $T ./blas_ref $s 50 2>&1 | tail -1
done
} > $OUT/r6b.txt 2>&1
cd ../..
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm or linear or nt" > gpurun_out/r6b/pytest_gemm.txt 2>&1
tail -3 gpurun_out/r6b/pytest_gemm.txt
