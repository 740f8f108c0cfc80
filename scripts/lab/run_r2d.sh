#!/bin/bash
# Round 2, sweep D: row-per-lane epilogue (swapped MFMA operands; ASE_NT8_V=128, 192 = with the DMA in the MFMA block)
L=scripts/lab/gemm_lab
for s in "16384 1024 1024" "32768 1024 1024" "16384 1024 512" "32768 1024 320" "12288 1024 1408" "32768 512 64" "32768 512 1024" "8192 8192 8192"; do
  for a in "0 1" "2 0" "3 1"; do
    for v in 0 64 128 192; do
      echo -n "V$v: "; ASE_NT8_V=$v ASE_NT_TILE=256 timeout 60 $L nt $s 20 $a | tail -1 || echo "rc=$?"
    done
  done
done
echo "--- edge shapes V192"
for s in "300 320 64" "1000 576 128" "16000 960 192" "777 256 256" "4096 1408 1024"; do
  for a in "0 1" "2 0" "3 1" "0 0"; do
    ASE_NT8_V=192 ASE_NT_TILE=256 timeout 60 $L nt $s 5 $a | tail -1 || echo "rc=$?"
  done
done
echo "--- stamps V192"
for s in "16384 1024 1024" "32768 512 64"; do
  LAB_PROF=1 ASE_NT8_V=192 ASE_NT_TILE=256 timeout 60 $L nt $s 20 0 1
  LAB_PROF=1 ASE_NT8_V=192 ASE_NT_TILE=256 timeout 60 $L nt $s 20 2 0
  LAB_PROF=1 ASE_NT8_V=192 ASE_NT_TILE=256 timeout 60 $L nt $s 20 3 1
done
