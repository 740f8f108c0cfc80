#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
L=scripts/lab/gemm_lab
( for s in "16384 1024 1024" "32768 1024 1024" "16384 1024 512" "32768 1024 320" "12288 1024 1408" "32768 512 1024" "32768 512 64" "16128 1024 1024" "8192 8192 8192"; do
    for aux in 0 3; do
      echo "== $s aux=$aux"
      echo -n "nt8  : "; ASE_NT4R=0 LAB_PROF=1 timeout 60 $L nt $s 20 $aux 1 | tail -2 | tr '\n' ' '; echo
      echo -n "nt4r : "; ASE_NT4R=1 LAB_PROF=1 timeout 60 $L nt $s 20 $aux 1 | tail -2 | tr '\n' ' '; echo
    done
  done
) > $O/nt4r.log 2>&1
cat $O/nt4r.log
( echo "== nt4r ablations (wrong results by construction): 1 no loads, 3 no loads + no LDS writes, 4 no fragment reads, 7 MFMA + barriers only"
  for s in "16384 1024 1024" "8192 8192 8192"; do
    for a in 0 1 3 4 7; do echo -n "$s abl=$a: "; ASE_NT4R=1 ASE_NT4R_ABL=$a LAB_PROF=1 timeout 60 scripts/lab/gemm_lab nt $s 10 0 1 | tail -2 | tr '\n' ' '; echo; done
  done ) > gpurun_out/r3i/nt4r_abl.log 2>&1
cat gpurun_out/r3i/nt4r_abl.log
