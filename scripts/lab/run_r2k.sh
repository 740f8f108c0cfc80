#!/bin/bash
# Round 2, sweep K: the four-wave 256 x 256 kernel (v40: one wave per SIMD, 128 x 128 wave tiles, reads / DMA issued between
# the MFMAs of the same wave) against the phased eight-wave kernel (v0).  v44 / v48 / v52 / v56: no DMA / no reads / neither / no MFMAs.
L=scripts/lab/gemm_lab
echo "== correctness, short and odd contraction lengths, ragged rows"
for s in "512 256 64" "512 256 128" "768 512 192" "1000 256 256" "4096 1024 320" "16384 1024 1024"; do
  for a in "0 1" "2 0" "3 1"; do
    echo -n "v40 aux/relu $a: "; ASE_NT_VARIANT=40 timeout 60 $L nt $s 3 $a | tail -1 || echo "rc=$?"
  done
done
echo "== timing"
for s in "8192 8192 8192" "16384 1024 1024" "32768 1024 1024" "32768 1024 320" "12288 1024 1408" "16384 1024 512"; do
  for v in 0 40; do
    echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 20 2 0 | tail -1 || echo "rc=$?"
  done
done
echo "== ablations"
for s in "8192 8192 8192" "16384 1024 1024"; do
  for v in 44 48 52 56; do
    echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 20 2 0 | tail -1 | sed 's/maxerr.*//' || echo "rc=$?"
  done
done
echo "== phase timestamps"
for v in 0 40; do echo "v$v"; ASE_NT_VARIANT=$v LAB_PROF=1 timeout 60 $L nt 16384 1024 1024 5 2 0 | tail -8; done
