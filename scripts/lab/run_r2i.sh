#!/bin/bash
# Round 2, sweep I: row-per-lane epilogue on the lock-step tiles (ASE_NT_ROWS=1) against the LDS-slab epilogue (0)
L=scripts/lab/gemm_lab
for s in "12288 512 1024" "16384 512 1024" "4096 1024 1408" "4096 1408 1024" "4096 512 1024" "4096 1024 1024" "32768 256 512" "12288 512 128" "16384 512 64" "32768 256 64" "2048 1024 1024"; do
  for a in "0 1" "2 0" "3 1"; do
    for r in 0 1; do
      echo -n "rows $r: "; ASE_NT_ROWS=$r timeout 60 $L nt $s 20 $a | tail -1 || echo "rc=$?"
    done
  done
done
echo "--- edge shapes (rows 1)"
for s in "300 320 64" "1000 576 128" "16000 960 192" "777 256 256" "100 64 64" "5000 192 320"; do
  for a in "0 1" "2 0" "3 1" "0 0"; do
    timeout 60 $L nt $s 5 $a | tail -1 || echo "rc=$?"
  done
done
