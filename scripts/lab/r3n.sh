#!/bin/bash
O=gpurun_out/r3n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_agent.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --steps 20 --warmup 5 --verbose > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench rc=$?" >> $O/rc.txt
timeout 600 python bench.py --steps 8 --warmup 2 --verbose --modes f16 --cpu-steps 4 > $O/bench_b.json 2> $O/bench_b.err
tail -3 $O/tests.log; cat $O/rc.txt
python - <<'PY'
import json
for f in ('gpurun_out/r3n/bench_bf16.json','gpurun_out/r3n/bench_b.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], 'qual', d['qualifying_mode'] and (d['qualifying_mode']['precision'], d['qualifying_mode']['fresh_max_loss_rel']))
    for m,r in d['modes'].items():
        fr=r['parity']['fresh']; st=r['parity']['stress']
        print('  ',m, r['ms_per_step'], 'fresh', fr['max_loss_rel'], fr['max_loss_rel_scalar'], fr['loss_rel']['disc_grad_penalty'], 'stress', st['max_loss_rel'], st['loss_rel']['disc_grad_penalty'])
PY
