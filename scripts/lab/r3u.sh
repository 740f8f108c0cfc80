#!/bin/bash
O=gpurun_out/r3u; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
tail -3 $O/tests.log; cat $O/rc.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3u/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['sustained_clock_mhz'])
q=d['qualifying_mode']; print(q and {k:v for k,v in q.items() if k not in ('criterion','by_mode')}); print(q and q['by_mode'])
for m,r in d['modes'].items(): print(m, r['ms_per_step'], r['parity']['fresh']['max_loss_rel'], r['parity']['fresh']['max_loss_rel_scalar'], r['parity']['stress']['max_loss_rel'])
PY
