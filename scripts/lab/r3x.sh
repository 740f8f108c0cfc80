#!/bin/bash
O=gpurun_out/r3x; mkdir -p $O
python scripts/lab/gp_scale_ab.py f16gpx3 2>&1 | grep -v amdgpu | tail -4 > $O/gpx3_err.log
timeout 300 python bench.py --precision f16gpx3 --steps 10 --warmup 3 --no-cpu-baseline --verbose > $O/bench.json 2> $O/bench.err
cat $O/gpx3_err.log; grep -o '"ms_per_step": [0-9.]*' $O/bench.json | head -1
