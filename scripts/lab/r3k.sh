#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
( echo "== shader clock under load: 'epilogue+drain' x 1000 = shader clocks of the main loop (abl 8: full kernel, 15: MFMA + barriers only)"
  for s in "16384 1024 1024" "8192 8192 8192"; do
    for a in 8 15; do echo -n "$s abl=$a: "; ASE_NT4R=1 ASE_NT4R_ABL=$a LAB_PROF=1 timeout 60 scripts/lab/gemm_lab nt $s 10 0 1 | tail -2 | tr '\n' ' '; echo; done
  done
) > gpurun_out/r3k/nt4r_clk.log 2>&1
cat gpurun_out/r3k/nt4r_clk.log
