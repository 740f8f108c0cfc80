#!/bin/bash
# Round 2, sweep F: tile choice for the launches nt_choice still sends to the 128 x 128 kernel, against the (now faster)
# phased 256 x 256 kernel forced by ASE_NT_TILE=256
L=scripts/lab/gemm_lab
for s in "12288 512 1024" "16384 512 1024" "4096 1024 1408" "4096 1408 1024" "4096 512 1024" "4096 1024 1024" "32768 256 512" "12288 512 128" "16384 512 64" "32768 256 64" "2048 1024 1024" "2048 1024 320" "2048 512 1024" "1536 1024 1408"; do
  for a in "2 0" "3 1"; do
    for t in 0 256; do
      echo -n "tile $t: "; ASE_NT_TILE=$t timeout 60 $L nt $s 20 $a | tail -1 || echo "rc=$?"
    done
  done
done
