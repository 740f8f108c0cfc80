cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
L=scripts/lab/gemm_lab
( for s in "16384 1024 1024" "32768 1024 1024" "16384 1024 512" "32768 1024 320" "12288 1024 1408" "32768 512 1024" "8192 8192 8192"; do
    for aux in 3 2; do
      echo "== $s aux=$aux"
      echo -n "nt8  : "; LAB_PROF=1 timeout 60 $L nt $s 20 $aux 1 | tail -2 | tr '\n' ' '; echo
      echo -n "nt8p : "; LAB_PACK=1 LAB_PROF=1 timeout 60 $L nt $s 20 $aux 1 | tail -2 | tr '\n' ' '; echo
    done
  done
  echo "== ablations on 16384 1024 1024 (wrong results by construction)"
  for v in 320 328 4 8 16; do echo -n "v$v: "; ASE_NT8_V=$v timeout 60 $L nt 16384 1024 1024 20 3 1 | tail -1; done
  for v in 320 328; do echo -n "8192^3 v$v: "; ASE_NT8_V=$v timeout 60 $L nt 8192 8192 8192 10 3 1 | tail -1; done
) > $O/nt8p.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
timeout 300 python bench.py --steps 10 --warmup 3 --precision bf16 --no-cpu-baseline --breakdown > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 300 python bench.py --steps 10 --warmup 3 --precision f16 --no-cpu-baseline > $O/bench_f16.json 2> $O/bench_f16.err
tail -5 $O/pytest.log
