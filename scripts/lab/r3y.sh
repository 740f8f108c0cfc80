#!/bin/bash
O=gpurun_out/r3y; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -k "f16gpx3" > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
tail -2 $O/tests.log; cat $O/rc.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3y/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
q=d['qualifying_mode']; print({k:v for k,v in q.items() if k not in ('criterion','by_mode')})
for m,r in d['modes'].items(): print(m, r['value'], r['ms_per_step'], r['parity']['fresh']['max_loss_rel'], r['parity']['fresh']['max_loss_rel_scalar'], r['parity']['fresh']['loss_rel']['disc_grad_penalty'], r['parity']['stress']['max_loss_rel'])
PY
