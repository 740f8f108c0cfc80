#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w; mkdir -p $O
L=scripts/lab/gemm_lab
( for s in "32768 64 1024" "32768 64 512" "16384 64 512" "32768 64 256" "12288 128 512" "32768 256 512"; do
    for v in 0 50 51 52 53; do
      echo -n "$s variant=$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 30 0 0 | tail -1
    done
  done ) > $O/skinny.log 2>&1
cat $O/skinny.log
