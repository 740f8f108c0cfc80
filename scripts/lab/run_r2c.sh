#!/bin/bash
# Round 2, sweep C: DMA issued inside the MFMA block (ASE_NT8_V=64) against the production schedule (0)
L=scripts/lab/gemm_lab
for s in "8192 8192 8192" "4096 4096 4096" "16384 1024 1024" "32768 1024 1024" "16384 1024 512" "32768 1024 320" "12288 1024 1408" "32768 512 64" "131072 1024 1024"; do
  for v in 0 64 66 72 80; do
    echo -n "V$v: "; ASE_NT8_V=$v ASE_NT_TILE=256 timeout 60 $L nt $s 20 0 1 | tail -1 || echo "rc=$?"
  done
done
echo "--- correctness of V64 on edge shapes and aux modes"
for s in "300 260 64" "1000 520 128" "16000 1000 192" "777 256 256" "4096 1408 1024" "16384 1024 1024" "12288 1024 1408"; do
  for a in "1 1" "2 0" "3 1"; do
    ASE_NT8_V=64 ASE_NT_TILE=256 timeout 60 $L nt $s 5 $a | tail -1 || echo "rc=$?"
  done
done
