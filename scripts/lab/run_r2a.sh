#!/bin/bash
# Round 2, sweep A: co-resident NT tilings (ASE_NT_VARIANT 10..17, <= 80 KB of LDS => two workgroups per CU) against the
# production choice (variant 0 = nt_choice: phased 256 x 256 or 128 x 128) on the in-situ shapes of one optimisation step.
# args of gemm_lab: nt M N K reps aux relu   (aux 0 none / 2 bit-mask consumer / 3 mask_out producer)
L=scripts/lab/gemm_lab
run() { # shape aux relu
  for v in 0 10 11 12 13 14 15 16 17; do
    echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $1 $2 $3 20 $4 $5 || echo "rc=$?"
  done
}
echo "=== forward (relu + mask_out)"
for s in "16384 1024 1024" "32768 1024 1024" "32768 1024 320" "12288 1024 1408" "32768 512 1024" "12288 512 1024"; do run $s 3 1; done
echo "=== data gradient (bit mask consumer)"
for s in "16384 1024 1024" "32768 1024 1024" "16384 1024 512" "32768 1024 512" "32768 512 64" "32768 512 256" "12288 512 128" "4096 1024 1408"; do run $s 2 0; done
echo "=== plain, long"
for s in "131072 1024 1024" "8192 8192 8192"; do run $s 0 1; done
