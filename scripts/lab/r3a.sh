set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3a/pytest.log
timeout 600 python bench.py --steps 6 --warmup 2 --precision f16 --cpu-steps 4 --no-parity-mode > gpurun_out/r3a/bench_f16.json 2> gpurun_out/r3a/bench_f16.err
timeout 600 python bench.py --steps 6 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/r3a/bench_bf16.json 2> gpurun_out/r3a/bench_bf16.err
tail -3 gpurun_out/r3a/pytest.log
