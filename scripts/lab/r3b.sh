set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r3b/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 --breakdown > gpurun_out/r3b/bench_bf16.json 2> gpurun_out/r3b/bench_bf16.err
timeout 300 python bench.py --steps 20 --warmup 5 --precision f16 --no-cpu-baseline > gpurun_out/r3b/bench_f16.json 2> gpurun_out/r3b/bench_f16.err
timeout 300 python bench.py --steps 20 --warmup 5 --precision bf16 --no-cpu-baseline > gpurun_out/r3b/bench_bf16_b.json 2> gpurun_out/r3b/bench_bf16_b.err
tail -5 gpurun_out/r3b/pytest.log
