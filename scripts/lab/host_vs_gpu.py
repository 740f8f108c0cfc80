"""Host submission time against GPU time of one rank's share of the sharded update (scripts/bench_extra.py --shard-of R): the update() call
returns when every launch-program entry of its 48 steps has been submitted; torch.cuda.synchronize() behind it waits for the GPU.
host ~ total means the step is bound by the replay loop of the launch program, not by its kernels."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import ase_amd  # noqa: E402
ase_amd.configure(cpu_threads=1)
import torch  # noqa: E402
import bench_extra  # noqa: E402

for prec in sys.argv[1].split(','):
    for R in [int(x) for x in sys.argv[2].split(',')]:
        ag, cfg, spec = bench_extra.build('ase', 4096, prec, overrides={'minibatch_size': 16384 // R, 'amp_minibatch_size': 4096 // R})
        one = lambda: ag.update(ag._play_steps_tail(), max_steps=48)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        host, total = [], []
        for _ in range(6):
            t0 = time.perf_counter()
            one()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host.append(t1 - t0)
            total.append(t2 - t0)
        progs = [g for g in ag._graphs.values() if not g['hipgraph']]
        entries = max((sum(ag.backend.prog_size(p) for p in g['graphs']) for g in progs), default=0)
        h, t = sorted(host)[len(host) // 2], sorted(total)[len(total) // 2]
        print(json.dumps({'precision': prec, 'ranks': R, 'entries_per_step': entries, 'host_ms_per_update': round(h * 1e3, 2),
                          'total_ms_per_update': round(t * 1e3, 2), 'host_us_per_step': round(h * 1e6 / 48, 1),
                          'total_us_per_step': round(t * 1e6 / 48, 1), 'host_us_per_entry': round(h * 1e6 / 48 / max(entries, 1), 2)}), flush=True)
        del ag
        torch.cuda.empty_cache()
