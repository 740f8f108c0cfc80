#!/bin/bash
# Lab ablation (timing only): the phased NT kernel's loop with its 8 x 32x32x16 MFMAs per phase (ASE_NT8_V=64, the product's
# schedule) against the same flop count as 16 x 16x16x32 MFMAs on eight independent accumulators (ASE_NT8_V=576; results are
# wrong by construction - "bad" in the check column is expected).   make -C scripts/lab gemm_lab && bash scripts/lab/mfma_shape.sh
cd "$(dirname "$0")"
for s in "16384 1024 1024" "32768 1024 1024" "4096 4096 4096" "8192 8192 8192"; do
  for rep in 1 2; do
    for v in 64 576; do
      echo "V=$v  $(ASE_NT8_V=$v ./gemm_lab nt $s 30 0 1 2>&1 | grep -E 'TF' | head -1)"
    done
  done
done
