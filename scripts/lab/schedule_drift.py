"""How far two SCHEDULES of the same update drift apart (the quantities tests/test_gpu_engine.py::test_schedule_variants_agree_config2
bounds), printed instead of asserted: serial schedule against the default one and against the default one without result rings, three
updates under program replay, n repetitions.     python scripts/lab/schedule_drift.py [precision] [repetitions]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import torch

import bench

precision = sys.argv[1] if len(sys.argv) > 1 else 'f32'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
KEYS = ('kl', 'actor_loss', 'critic_loss', 'disc_loss', 'disc_grad_penalty', 'enc_loss', 'amp_diversity_loss')
for rep in range(reps):
    outs = []
    for opts, extra in (({'xstep': False, 'prefetch': False, 'gp_stream': False}, {'main_stream_priority': 0, 'result_rings': False}),
                        ({}, {}), ({}, {'result_rings': False})):
        agent, cfg, spec = bench.make_agent('cuda:0', precision, 'program', 1, 0, engine_opts=opts or {'xstep': True}, extra_cfg=extra)
        bench.fill_rollout(agent, 'cuda:0')
        agent._init_amp_demo_buf()
        infos = [agent.update(agent._play_steps_tail()) for _ in range(3)]
        torch.cuda.synchronize()
        outs.append((agent.model.a2c_network.flat_params.detach().float().cpu().clone(), agent.engine.obs_state.cpu().clone(),
                     agent.engine.amp_state.cpu().clone(),
                     {k: torch.stack([torch.as_tensor(x).float().reshape(-1)[0].cpu() for x in infos[-1][k]]) for k in KEYS}))
        del agent
        torch.cuda.empty_cache()
    lr = 2e-5
    for name, (w1, o1, a1, r1) in (('default', outs[1]), ('no rings', outs[2])):
        w0, o0, a0, r0 = outs[0]
        sc = {k: abs(float(r1[k].mean()) - float(r0[k].mean())) / max(abs(float(r0[k].mean())), 0.05) for k in KEYS}
        worst = max(sc, key=sc.get)
        print(f'{precision} rep {rep} serial vs {name}: weights max {float((w0 - w1).abs().max()) / lr:.1f} lr mean {float((w0 - w1).abs().mean()) / lr:.3f} lr; '
              f'obs state {float((o0 - o1).abs().max()):.2e} amp state {float((a0 - a1).abs().max()):.2e}; worst scalar {worst} {sc[worst]:.2e}; '
              f'finite {all(bool(torch.isfinite(r1[k]).all()) for k in KEYS)}', flush=True)
