#!/bin/bash
# Round 6, GPU call Z: kernel traces of the throughput mode (bf16) on the final build, and of an update under mixed_precision: True
# (what the dynamic loss scale launches per step now: no check pass over the half buffers).
cd "$(dirname "$0")/../.."
O=gpurun_out/r6z; mkdir -p $O
sha256sum ase_amd/csrc/libase_hip.so > $O/lib_sha256.txt
SKIP_PMC=1 timeout 900 bash scripts/profile_round.sh bf16 > $O/profile_bf16.log 2>&1
cp gpurun_out/profile_bf16/kernel_stats_serial.txt $O/kernel_stats_bf16_serial.txt
cp gpurun_out/profile_bf16/kernel_stats_replay.txt $O/kernel_stats_bf16_replay.txt
cp gpurun_out/profile_bf16/timeline_replay.txt $O/timeline_replay_bf16.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=/tmp/prof_r6z; rm -rf $R
rocprofv3 --kernel-trace --stats -d $R/m -o t -- python scripts/bench_extra.py --only ase-mixed --updates 4 > $O/mixed.log 2>&1
python scripts/rocpd_stats.py $R/m/t_results.db 60 > $O/kernel_stats_mixed_replay.txt
rm -rf $R
grep -h "ase-mixed" $O/mixed.log | cut -c1-300
grep -i "scaler" $O/kernel_stats_mixed_replay.txt
