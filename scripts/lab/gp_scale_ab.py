"""Lab (usage: gp_scale_ab.py [f16|f16gp32]): f16 parity of the reported gradient penalty with the gradient scale split between the chain's two factors
(engine_opts['gp_scale_split'] = True) and with all of it on the dJ/dU side (False): same weights, statistics, minibatch -
the running statistics are restored between the two engine steps, the f32 CPU oracle (oracle/restated.py) is evaluated once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ase_amd
ase_amd.configure(cpu_threads=1)
import torch
import bench
from oracle import restated as R

dev = 'cuda:0'
MODE = sys.argv[1] if len(sys.argv) > 1 else 'f16'
agent, cfg, _ = bench.make_agent(dev, MODE, 'program', 1, 0)
bench.fill_rollout(agent, dev)
agent._init_amp_demo_buf()
eng = agent.engine
B, MB, AMB = agent.batch_size, agent.minibatch_size, cfg['amp_minibatch_size']
for rnd in range(4):
    for _ in range(7):
        agent.update(agent._play_steps_tail())
    bench.fill_rollout(agent, dev)
    agent._play_steps_tail()
    torch.cuda.synchronize()
    H, N = agent._remap
    env_major = lambda t: t.view(H, N, -1).transpose(0, 1).reshape(H * N, -1)
    ds = {k: env_major(v).cpu() for k, v in agent._ds.items()}
    for k in ('old_logp_actions', 'advantages', 'rand_action_mask'):
        ds[k] = ds[k].view(-1)
    ds['amp_obs_replay'] = ds['amp_obs']
    g = torch.Generator().manual_seed(rnd)
    demo = agent._amp_obs_demo_buffer.data
    dsel = torch.randint(0, demo.shape[0], (B,), generator=g)
    ds['amp_obs_demo'] = demo.cpu()[dsel]
    sd = R.canonical_sd(agent.model.state_dict(), False,
                        requires_grad=[k.replace('a2c_network.', '', 1) for k, p in agent.model.named_parameters() if p.requires_grad])
    sd = {k: (v.cpu().detach().requires_grad_(True) if v.requires_grad else v.cpu()) for k, v in sd.items()}
    rms = {'obs': bench._rms_dict(eng.obs_state), 'amp': bench._rms_dict(eng.amp_state)}
    idx = torch.randperm(B, generator=g)[:MB]
    mb = {k: v[idx] for k, v in ds.items()}
    z = R.sample_latents(MB, 64, g)
    torch.set_num_threads(16)
    ref = R.calc_gradients('ase', sd, rms, mb, cfg, z)
    torch.set_num_threads(1)
    idx_d = idx.to(torch.int32).to(dev)
    arows = idx_d[:AMB].contiguous()
    streams = [(agent._ds['amp_obs'], arows, agent._remap), (agent._ds['amp_obs'], arows, agent._remap),
               (demo, dsel[idx[:AMB]].to(torch.int32).to(dev), (0, 0))]
    keep = (eng.obs_state.clone(), eng.amp_state.clone())
    r = float(ref['disc_grad_penalty'].mean())
    out = []
    for split in ((True, False, True, False) if MODE == 'f16' else (True,)):
        eng.engine_opts['gp_scale_split'] = split
        eng.obs_state.copy_(keep[0]); eng.amp_state.copy_(keep[1])
        eng.step(agent._ds, idx_d, agent._remap, streams, new_z=z.to(dev), apply=False)
        torch.cuda.synchronize()
        v = float(eng.results()['disc_grad_penalty'].mean())
        out.append(f'split={split}: {(v - r) / r:+.2e}')
    eng.obs_state.copy_(keep[0]); eng.amp_state.copy_(keep[1])
    eng.engine_opts['gp_scale_split'] = True
    print(f'[{MODE}] after {7 * (rnd + 1)} updates: penalty {r:.5f} |', ' | '.join(out), flush=True)
