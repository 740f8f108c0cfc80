#!/bin/bash
# Round 2, sweep E: production default (DMA in the MFMA block + row-per-lane epilogue, mask words by DMA) vs round-1 schedule
L=scripts/lab/gemm_lab
for s in "16384 1024 1024" "32768 1024 320" "12288 1024 1408" "32768 512 64" "16384 1024 512"; do
  for a in "0 1" "2 0" "3 1"; do
    for v in 0 -2; do
      echo -n "V$v: "; ASE_NT8_V=$v ASE_NT_TILE=256 timeout 60 $L nt $s 20 $a | tail -1 || echo "rc=$?"
    done
  done
done
echo "--- edge shapes, default"
for s in "300 320 64" "1000 576 128" "16000 960 192" "777 256 256" "4096 1408 1024" "300 260 64" "1000 520 128"; do
  for a in "0 1" "2 0" "3 1" "1 0"; do
    ASE_NT_TILE=256 timeout 60 $L nt $s 5 $a | tail -1 || echo "rc=$?"
  done
done
echo "--- stamps"
LAB_PROF=1 ASE_NT_TILE=256 timeout 60 $L nt 16384 1024 1024 20 2 0
LAB_PROF=1 ASE_NT_TILE=256 timeout 60 $L nt 32768 512 64 20 2 0
