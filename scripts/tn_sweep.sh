for t in 256 512 1024 2048; do echo "== ASE_TN_TARGET_WG=$t"; ASE_TN_TARGET_WG=$t python scripts/gemm_bench.py 2>&1 | grep "^TN"; done
