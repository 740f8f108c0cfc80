#!/bin/bash
O=gpurun_out/r3s; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -q -k self_consistent > $O/tests.log 2>&1; echo "tests rc=$?" > $O/rc.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/rc.txt
tail -2 $O/tests.log; cat $O/rc.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3s/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['sustained_clock_mhz'])
print(d['qualifying_mode'] and {k:v for k,v in d['qualifying_mode'].items() if k not in ('criterion',)})
for m,r in d['modes'].items(): print(m, r['ms_per_step'], r['timed'])
PY
