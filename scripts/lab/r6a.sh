#!/bin/bash
# Round 6, GPU call A: (1) LDS-DMA issue micro-benchmark, (2) loop rate / fixed cost of the product's phased NT kernel and of the vendor
# kernel from a K sweep, (3) the 4-wave half-K-tile-pipelined lab kernel (gemm_nt4v_variant.hip) with its ablations.
cd "$(dirname "$0")"
OUT=../../gpurun_out/r6a; mkdir -p $OUT
T="timeout 120"
{
echo "#### dma_bench"; $T ./dma_bench
echo "#### K sweep, 16384 x 1024 x K: product nt8 (LAB_PROF) | rocBLAS"
for K in 512 1024 2048 4096; do
  LAB_PROF=1 $T ./gemm_lab nt 16384 1024 $K 50 0 1 2>&1 | grep -E "profile|TF"
  $T ./blas_ref 16384 1024 $K 50 2>&1 | tail -1
done
echo "#### nt4v variants (ASE_NT4V=v): correctness + time, 16384 / 32768 x 1024 x 1024"
for v in 2 0 3 6 10 14; do
  for M in 16384 32768; do
    echo "-- ASE_NT4V=$v M=$M"; ASE_NT4V=$v LAB_PROF=1 $T ./gemm_lab nt $M 1024 1024 50 0 1 2>&1 | grep -E "profile|TF|mismatch|failed" | head -6
  done
done
echo "#### nt4v shader clocks of the loop (x1000 in the epilogue column): full / MFMA only"
for v in 18 30; do ASE_NT4V=$v LAB_PROF=1 $T ./gemm_lab nt 16384 1024 1024 30 0 1 2>&1 | grep -E "profile|TF" ; done
echo "#### nt4v masks: aux=2 (consume bit mask), aux=3 (produce), ragged M"
for v in 2 0; do for aux in 2 3; do ASE_NT4V=$v $T ./gemm_lab nt 16384 1024 1024 20 $aux 1 2>&1 | grep -E "TF|mismatch" | head -4; done; done
ASE_NT4V=2 $T ./gemm_lab nt 16300 1024 1024 20 3 1 2>&1 | grep -E "TF|mismatch" | head -4
ASE_NT4V=2 $T ./gemm_lab nt 12288 1024 1408 20 0 1 2>&1 | grep -E "TF|mismatch" | head -4
ASE_NT4V=2 $T ./gemm_lab nt 16384 1024 128 20 0 1 2>&1 | grep -E "TF|mismatch" | head -4
ASE_NT4V=2 $T ./gemm_lab nt 16384 1024 64 20 0 1 2>&1 | grep -E "TF|mismatch" | head -4
echo "#### K sweep nt4v (v=2, v=0)"
for v in 2 0; do for K in 512 1024 2048 4096; do ASE_NT4V=$v LAB_PROF=1 $T ./gemm_lab nt 16384 1024 $K 50 0 1 2>&1 | grep -E "profile|TF"; done; done
echo "#### other carrying shapes: product | nt4v v=2 | v=0 | rocBLAS"
for s in "32768 1024 320" "16384 1024 512" "32768 512 1024" "12288 1024 1408" "131072 1024 1024"; do
  echo "== $s"; $T ./gemm_lab nt $s 30 0 1 2>&1 | grep TF; ASE_NT4V=2 $T ./gemm_lab nt $s 30 0 1 2>&1 | grep TF; ASE_NT4V=0 $T ./gemm_lab nt $s 30 0 1 2>&1 | grep TF; $T ./blas_ref $s 30 2>&1 | tail -1
done
} > $OUT/r6a.txt 2>&1
tail -5 $OUT/r6a.txt
