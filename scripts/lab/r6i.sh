#!/bin/bash
# Round 6, GPU call I: the round's measurement set on the FINAL build (ABI 7) - one box for all of it:
#   (1) the driver-form bench line, (2) kernel traces + PMC passes of the same command (scripts/profile_round.sh), (3) the same update
#   replayed from captured hipGraphs, (4) the side measurements (mixed_precision, the headline mode under the dynamic scale, the other
#   configurations, one rank's share of the sharded update), (5) the GPU suite.
cd "$(dirname "$0")/../.."
O=gpurun_out/r6i; mkdir -p $O
sha256sum ase_amd/csrc/libase_hip.so > $O/lib_sha256.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $O/bench_n1_detail.json > $O/bench_n1.json 2> $O/bench_n1.err
tail -1 $O/bench_n1.json | cut -c1-900
timeout 1500 bash scripts/profile_round.sh f16gpx3 > $O/profile_round.log 2>&1
cp -r gpurun_out/profile_f16gpx3 $O/
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --precision f16gpx3 --hipgraph --no-cpu-baseline --no-config5 --throughput-mode none --detail '' > $O/bench_hipgraph.json 2> $O/bench_hipgraph.err
tail -1 $O/bench_hipgraph.json | cut -c1-300
timeout 900 python scripts/bench_extra.py --updates 8 > $O/bench_extra.jsonl 2> $O/bench_extra.err
cut -c1-260 $O/bench_extra.jsonl
timeout 600 python scripts/bench_extra.py --shard-of 2,4,8 --precision f16gpx3 --updates 6 > $O/shard_compute.jsonl 2> $O/shard_compute.err
timeout 600 python scripts/bench_extra.py --shard-of 2,4,8 --precision bf16 --updates 6 >> $O/shard_compute.jsonl 2>> $O/shard_compute.err
cut -c1-260 $O/shard_compute.jsonl
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -2 $O/pytest_gpu.txt
