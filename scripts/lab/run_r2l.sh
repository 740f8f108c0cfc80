#!/bin/bash
# Round 2, sweep L: issue patterns of the four-wave kernel (v40 reads first / v41 spread / v43 spread + odd waves one slot
# later; v44/45 no DMA, v48/49 no reads, v52 neither) against the phased eight-wave kernel (v0) and rocBLAS.
L=scripts/lab/gemm_lab
echo "== correctness of the new patterns"
for s in "512 256 64" "768 512 192" "4096 1024 320" "16384 1024 1024"; do
  for v in 41 43; do
    echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 3 2 0 | tail -1 || echo "rc=$?"
  done
done
echo "== timing"
for s in "8192 8192 8192" "16384 1024 1024" "32768 1024 1024" "32768 1024 320"; do
  for v in 0 40 41 43; do
    echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 20 2 0 | tail -1 | sed 's/maxerr.*bad/bad/' || echo "rc=$?"
  done
done
echo "== ablations 8192^3"
for v in 44 45 48 49 52; do
  echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt 8192 8192 8192 20 2 0 | tail -1 | sed 's/maxerr.*//' || echo "rc=$?"
done
echo "== phase timestamps"
for v in 40 41; do echo "v$v"; ASE_NT_VARIANT=$v LAB_PROF=1 timeout 60 $L nt 16384 1024 1024 5 2 0 | tail -3 | head -1; done
echo "== rocBLAS"
for s in "8192 8192 8192" "16384 1024 1024" "32768 1024 1024" "32768 1024 320" "12288 1024 1408" "16384 512 1024" "4096 1024 1024"; do
  timeout 120 scripts/lab/blas_ref $s 20 || echo "rc=$?"
done
