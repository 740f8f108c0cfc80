#!/bin/bash
# Round profile of the benchmark command on the GPU box (run through gpurun from the repo root):
#   kernel-trace summaries (default replay run + a serial eager run whose per-kernel averages are comparable with bench.py's
#   HIP-event roofline pass) and the PMC passes (one counter group per run, --kernel-trace only).
# Output: gpurun_out/profile[_<precision>]/*.txt  (copy what is to be kept into profiles/).
#   bash scripts/profile_round.sh [precision]      (default: bench.py's headline mode)
PREC=${1:+--precision $1}
OUT=gpurun_out/profile${1:+_$1}; mkdir -p $OUT; rm -f $OUT/pmc_summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=/tmp/prof_round; rm -rf $R
rocprofv3 --kernel-trace --stats -d $R/g -o t -- python bench.py --no-cpu-baseline --no-config5 --throughput-mode none $PREC > $OUT/bench_replay.log 2>&1
python scripts/rocpd_stats.py $R/g/t_results.db 40 > $OUT/kernel_stats_replay.txt
python scripts/rocpd_timeline.py $R/g/t_results.db 60 > $OUT/timeline_replay.txt
rm -rf $R/g
SER="python bench.py --no-graph --no-multi-stream --steps 2 --warmup 1 --no-cpu-baseline --no-config5 --throughput-mode none $PREC"
rocprofv3 --kernel-trace --stats -d $R/s -o t -- $SER > $OUT/bench_serial.log 2>&1
python scripts/rocpd_stats.py $R/s/t_results.db 40 > $OUT/kernel_stats_serial.txt
rm -rf $R/s
PMC="python bench.py --no-graph --no-multi-stream --steps 1 --warmup 0 --no-cpu-baseline --no-config5 --throughput-mode none $PREC"
i=0
[ -n "$SKIP_PMC" ] && { grep -h "^{\"metric\"" $OUT/bench_replay.log > $OUT/bench_replay.json; exit 0; }
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace -d $R/p$i -o t -- $PMC > /dev/null 2>&1
  echo "## $c" >> $OUT/pmc_summary.txt
  python scripts/pmc_summary.py $R/p$i/t_results.db | head -44 >> $OUT/pmc_summary.txt
  echo >> $OUT/pmc_summary.txt
  rm -rf $R/p$i
done
grep -h "^{\"metric\"" $OUT/bench_replay.log > $OUT/bench_replay.json
grep -h "^{\"metric\"" $OUT/bench_serial.log > $OUT/bench_serial.json
# the per-launch HBM traffic file bench.py quotes (copy to profiles/rNN_pmc.json together with the summaries)
python scripts/make_pmc_json.py $OUT/pmc_summary.txt $OUT/pmc.json $OUT/kernel_stats_serial.txt --lib ase_amd/csrc/libase_hip.so > $OUT/pmc_json.log 2>&1
sha256sum ase_amd/csrc/libase_hip.so > $OUT/lib_sha256.txt
