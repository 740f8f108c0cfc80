#!/bin/bash
# Round 6, GPU call AC: the measurement set again after the one schedule default that moved (engine_opts gp_stream: off) - same library
# as call I (the PMC passes and serial traces of that call are kernel-level and stand): GPU suite, driver-form bench line, replay trace +
# timeline of the headline mode, side measurements, one rank's share.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6ac; mkdir -p $O
sha256sum ase_amd/csrc/libase_hip.so > $O/lib_sha256.txt
timeout 1200 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_n1_detail.json > $O/bench_n1.json 2> $O/bench_n1.err
tail -1 $O/bench_n1.json | cut -c1-700
SKIP_PMC=1 timeout 900 bash scripts/profile_round.sh f16gpx3 > $O/profile_round.log 2>&1
cp gpurun_out/profile_f16gpx3/kernel_stats_replay.txt $O/kernel_stats_f16gpx3_replay.txt
cp gpurun_out/profile_f16gpx3/timeline_replay.txt $O/timeline_replay_f16gpx3.txt
cp gpurun_out/profile_f16gpx3/bench_replay.json $O/bench_profiled_replay.json
timeout 900 python scripts/bench_extra.py --updates 8 > $O/bench_extra.jsonl 2> $O/bench_extra.err
cut -c1-200 $O/bench_extra.jsonl
timeout 600 python scripts/bench_extra.py --shard-of 2,4,8 --precision f16gpx3 --updates 6 > $O/shard_compute.jsonl 2> $O/shard_compute.err
timeout 600 python scripts/bench_extra.py --shard-of 2,4,8 --precision bf16 --updates 6 >> $O/shard_compute.jsonl 2>> $O/shard_compute.err
cut -c1-330 $O/shard_compute.jsonl | grep -o '"ranks": [0-9]*, "ms_per_update": [0-9.]*, "us_per_step": [0-9.]*'
