#!/bin/bash
# Round 2, sweep M: is the LDS-DMA slow because of the XOR chunk swizzle on its SOURCE addresses (lanes of a quad no longer
# ascending inside their 64 bytes)?  v104 / v112 = v40 / v48 with linear sources (wrong results, timing only); rocBLAS on
# the same random operands.
L=scripts/lab/gemm_lab
for s in "8192 8192 8192" "16384 1024 1024"; do
  for v in 40 104 48 112 52; do
    echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 20 2 0 | tail -1 | sed 's/maxerr.*//' || echo "rc=$?"
  done
done
echo "== rocBLAS, random operands"
for s in "8192 8192 8192" "16384 1024 1024" "32768 1024 1024" "32768 1024 320" "12288 1024 1408" "16384 512 1024" "4096 1024 1024"; do
  timeout 120 scripts/lab/blas_ref $s 20 || echo "rc=$?"
done
echo "== rocBLAS kernel names"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/rb -- $GRAFT_REPO_ROOT/scripts/lab/blas_ref 16384 1024 1024 5 > /dev/null 2>&1; find /tmp/rb -name "*kernel_stats*" | head -1 | xargs head -5 | cut -c1-300
