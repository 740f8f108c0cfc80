#!/bin/bash
# Round 6, GPU call D: (1) the GPU suite on the ABI-6 build (loss scale on the device), (2) config 2 under mixed_precision: True -
# skipped steps from GradScaler's initial 65536, (3) chip power / clocks while the benchmark update runs, (4) same-box A/B of the NT
# dispatch policy: p0 = 8-wave phased kernel everywhere, libase_hip.so = 4-wave kernel for the wide launches, p2 = the same under a
# 464-register cap (48 registers per SIMD left for co-resident small kernels).
cd "$(dirname "$0")/../.."
O=gpurun_out/r6d; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
timeout 600 python scripts/bench_extra.py --only ase-mixed --updates 8 > $O/bench_mixed.jsonl 2> $O/bench_mixed.err
tail -2 $O/bench_mixed.jsonl | cut -c1-600
# power / clock samples beside a 30-update run of the headline mode
( for i in $(seq 1 200); do rocm-smi --showpower --showclocks --csv 2>/dev/null | tr '\n' ' '; echo; sleep 0.2; done ) > $O/smi_samples.txt 2>&1 &
SMI=$!
timeout 600 python bench.py --gpus 1 --steps 30 --warmup 5 --precision f16gpx3 --no-cpu-baseline --no-config5 --throughput-mode none --detail '' > $O/bench_power_run.json 2> $O/bench_power_run.err
kill $SMI 2>/dev/null
REPS=2 timeout 2400 bash scripts/lab/ab_lib.sh libase_hip_p0.so libase_hip.so libase_hip_p2.so f16gpx3 > $O/ab_policy_f16gpx3.txt 2>&1
grep update $O/ab_policy_f16gpx3.txt
