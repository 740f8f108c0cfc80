cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05
bash scripts/profile_round.sh > gpurun_out/profile_round.log 2>&1
SKIP_PMC=1 bash scripts/profile_round.sh bf16 > gpurun_out/profile_round_bf16.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r05/bench_n1_detail.json > gpurun_out/r05/bench_n1.json 2> gpurun_out/r05/bench_n1.err
tail -c 600 gpurun_out/r05/bench_n1.json
python scripts/bench_extra.py > gpurun_out/r05/bench_extra.jsonl 2>/dev/null
python scripts/bench_extra.py --shard-of 2,4,8 --precision f16gpx3 > gpurun_out/r05/shard_compute.jsonl 2>/dev/null
python scripts/bench_extra.py --shard-of 2,4,8 --precision bf16 >> gpurun_out/r05/shard_compute.jsonl 2>/dev/null
ls gpurun_out/profile gpurun_out/profile_bf16
