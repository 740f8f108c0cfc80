set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
L=scripts/lab/gemm_lab
( for ws in 0 1; do echo "== tng ws=$ws"; LAB_TNG_WS=$ws LAB_PROF=1 timeout 120 $L tng 10; done ) > gpurun_out/r3c/tng.log 2>&1
( for s in "32768 64 1024" "32768 64 512" "32768 64 256" "16384 64 512"; do
    for v in 0 23 21 50 51 52 53; do echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 20 0 0 | tail -1; done; done
  for s in "32768 512 64" "32768 256 64" "16384 512 64" "12288 512 128"; do
    for v in 0 14 20 21 23; do echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 20 3 1 | tail -1; done; done
  for s in "12288 128 512" "4096 512 1024" "4096 1024 1024" "16384 512 1024" "12288 512 1024"; do
    for v in 0 14 20 21 22 16; do echo -n "v$v: "; ASE_NT_VARIANT=$v timeout 60 $L nt $s 20 3 1 | tail -1; done; done
) > gpurun_out/r3c/narrow.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r3c/pytest.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --precision bf16 --no-cpu-baseline --breakdown > gpurun_out/r3c/bench_bf16.json 2> gpurun_out/r3c/bench_bf16.err
tail -5 gpurun_out/r3c/pytest.log
