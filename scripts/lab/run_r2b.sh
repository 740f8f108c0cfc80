#!/bin/bash
# Round 2, sweep B: ablations of the phased NT kernel (ASE_NT8_V bit mask: 4 no DMA in the loop, 8 no fragment reads,
# 16 no MFMAs, 32 no epilogue, 1 / 2 schedule variants) - timing only, results of the ablated builds are wrong by design.
L=scripts/lab/gemm_lab
for s in "8192 8192 8192" "16384 1024 1024" "32768 512 64"; do
  for v in 0 4 8 12 16 28 32 2 1; do
    echo -n "V$v: "; ASE_NT8_V=$v ASE_NT_TILE=256 timeout 60 $L nt $s 20 0 1 | tail -1 || echo "rc=$?"
  done
done
echo "--- in-kernel stamps (V0)"
for s in "8192 8192 8192" "16384 1024 1024" "16384 1024 512" "32768 512 64"; do
  LAB_PROF=1 ASE_NT_TILE=256 timeout 60 $L nt $s 20 0 1
  LAB_PROF=1 ASE_NT_TILE=256 timeout 60 $L nt $s 20 2 0
  LAB_PROF=1 ASE_NT_TILE=256 timeout 60 $L nt $s 20 3 1
done
