"""Time the UNMODIFIED reference's optimisation step (ASEAgent.calc_gradients, /root/reference/ase/learning/ase_agent.py:159-308,
incl. loss.backward() and optimizer.step()) on the host cores, at BASELINE config 2's full size - and oracle/restated.py (the
travel-capable restatement bench.py times as `cpu_baseline`, kind "port") on the SAME inputs in the SAME process, so the
port's figure can be read as the reference's.  TEST / MEASUREMENT INFRASTRUCTURE ONLY: needs /root/reference (authoring
container); writes profiles/r04_reference_cpu_timing.json.

    python oracle/time_reference.py [--steps 6] [--threads N]
    python oracle/time_reference.py --config amp_cfg1      # BASELINE configs[0]: the reference's whole train_epoch update, CPU
"""
import argparse
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_runner  # noqa: E402
from oracle import restated as R  # noqa: E402
from oracle.make_golden import _yaml  # noqa: E402


def time_amp_cfg1(a):
    """BASELINE.json configs[0] ("this IS the baseline config"): the reference's AMPAgent - 64 envs x horizon 16,
    obs 253 / act 31 / amp obs 1400, [256, 128] MLPs, minibatch 256 / amp 64, 6 mini-epochs = 24 optimisation steps - running its
    OWN train_epoch (learning/amp_agent.py:181-264: demo refresh, tail of play_steps, dataset, 24 x calc_gradients + Adam, replay
    store) on the synthetic rollout, the simulator loop replaced by the statements behind it exactly as oracle/make_golden.py does
    for the golden vectors (amp_cfg1.pt).  Metric = BASELINE's: samples / update time."""
    from oracle import make_golden as MG
    from ase_amd.synthetic import EnvSpec, SyntheticSource
    ncpu = a.threads or len(os.sched_getaffinity(0))
    torch.set_num_threads(ncpu)
    net, cfg = MG._case('amp_cfg1')
    N, H = 64, cfg['horizon_length']
    spec = EnvSpec(num_envs=N, horizon=H, obs_size=253, act_size=31, amp_obs_size=1400, latent_dim=0, latent_steps_min=1,
                   latent_steps_max=2, episode_length=20)
    src = SyntheticSource(spec, seed=1234 + 21)
    torch.manual_seed(21)
    A = ref_runner.build_ref_agent('amp', net, cfg, num_envs=N, obs_size=253, act_size=31, amp_obs_size=1400,
                                   demo_fetch=src.fetch_amp_obs_demo, seed=21)
    A._init_amp_demo_buf()
    A.play_steps = lambda: MG._tail(A, 'amp')
    times = []
    for ep in range(a.steps + 2):
        exp = src.experience(MG._policy_from_agent(A, 'amp'), with_amp=True, with_latents=False)
        td = A.experience_buffer.tensor_dict
        for k, v in exp.items():
            if k in td:
                td[k].copy_(v)
        A.epoch_num += 1
        t0 = time.time()
        with ref_runner._Quiet():
            A.train_epoch()          # REFERENCE CODE, unmodified
        times.append(time.time() - t0)
    tt = sorted(times[2:])
    med = tt[len(tt) // 2]
    n_steps = cfg['mini_epochs'] * (N * H // cfg['minibatch_size'])
    out = {'what': 'BASELINE configs[0]: the UNMODIFIED reference AMPAgent.train_epoch (learning/amp_agent.py:181-264) without the simulator '
                   'loop, 64 envs x horizon 16 = 1024 samples, [256, 128] MLPs, 24 optimisation steps per update; median of the updates '
                   'after two warm-up ones, host CPU',
           'cores': ncpu, 'updates_timed': len(tt), 's_per_update': round(med, 4), 'samples_per_s': round(N * H / med, 1),
           'optimisation_steps_per_update': n_steps, 'ms_per_optimisation_step': round(med / n_steps * 1e3, 2),
           'all_s': [round(t, 4) for t in times], 'code': 'ase/learning/amp_agent.py (imported unmodified through oracle/ref_runner.py)'}
    path = os.path.join(ROOT, 'profiles', 'r05_reference_cpu_timing_cfg1.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='ase_cfg2', choices=['ase_cfg2', 'amp_cfg1'])
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--threads', type=int, default=0, help='torch CPU threads (0 = all cores this process may use)')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r04_reference_cpu_timing.json'))
    a = ap.parse_args()
    if a.config == 'amp_cfg1':
        return time_amp_cfg1(a)
    ncpu = a.threads or len(os.sched_getaffinity(0))
    torch.set_num_threads(ncpu)
    y = _yaml('ase_humanoid.yaml')                       # the reference's own yaml, verbatim: config 2
    net, cfg = y['network'], y['config']
    for k in ('learning_rate',):
        cfg[k] = float(cfg[k])
    N, H = 4096, cfg['horizon_length']
    MB, AMB = cfg['minibatch_size'], cfg['amp_minibatch_size']
    assert (N * H, MB, AMB) == (131072, 16384, 4096)
    A = ref_runner.build_ref_agent('ase', net, cfg, num_envs=N, obs_size=253, act_size=31, amp_obs_size=1400, seed=0)
    g = torch.Generator().manual_seed(1)

    def minibatch():
        z = torch.randn(MB, 64, generator=g)
        mu = torch.randn(MB, 31, generator=g) * 0.1
        return {'obs': torch.randn(MB, 253, generator=g), 'actions': mu + 0.055 * torch.randn(MB, 31, generator=g),
                'mu': mu, 'sigma': torch.full((MB, 31), 0.055023), 'old_logp_actions': torch.randn(MB, generator=g) * 3 - 40,
                'advantages': torch.randn(MB, generator=g), 'old_values': torch.randn(MB, 1, generator=g),
                'returns': torch.randn(MB, 1, generator=g), 'rand_action_mask': (torch.rand(MB, generator=g) < 0.8).float(),
                'ase_latents': z / z.norm(dim=-1, keepdim=True), 'amp_obs': torch.randn(MB, 1400, generator=g),
                'amp_obs_replay': torch.randn(MB, 1400, generator=g), 'amp_obs_demo': torch.randn(MB, 1400, generator=g)}

    mbs = [minibatch() for _ in range(a.steps + 1)]
    # ---- the reference itself
    t_ref = []
    for mb in mbs:
        t0 = time.time()
        with ref_runner._Quiet():
            A.calc_gradients(mb)
        t_ref.append(time.time() - t0)
    # ---- the restatement on the same inputs (fresh copy of the same initial weights is not needed for timing)
    sd = R.canonical_sd({k: v.detach().clone() for k, v in A.model.state_dict().items()}, False,
                        requires_grad=[k.replace('a2c_network.', '', 1) for k, p in A.model.named_parameters() if p.requires_grad])
    sd = {k: (v.detach().requires_grad_(True) if v.requires_grad else v) for k, v in sd.items()}
    rms = {'obs': R.rms_new(253), 'amp': R.rms_new(1400)}
    adam = R.adam_new()
    t_port = []
    for mb in mbs:
        t0 = time.time()
        R.calc_gradients('ase', sd, rms, mb, cfg, R.sample_latents(MB, 64, g))
        R.adam_step(sd, adam, cfg['learning_rate'])
        t_port.append(time.time() - t0)

    def med(t):
        t = sorted(t[1:])
        return t[len(t) // 2]
    n_steps = cfg['mini_epochs'] * (N * H // MB)
    out = {'what': 'one full-size optimisation step (minibatch 16384, amp 4096, 7,039,905 parameters) on the host: the unmodified '
                   'reference (ASEAgent.calc_gradients: forward, losses, backward, Adam) and oracle/restated.py on the same inputs, '
                   'same process; median of the steps after one warm-up',
           'cores': ncpu, 'steps': a.steps,
           'reference': {'s_per_step': round(med(t_ref), 3), 'samples_per_s': round(N * H / (n_steps * med(t_ref)), 1),
                         'all': [round(t, 3) for t in t_ref], 'code': 'ase/learning/ase_agent.py:159-308 (imported unmodified)'},
           'port': {'s_per_step': round(med(t_port), 3), 'samples_per_s': round(N * H / (n_steps * med(t_port)), 1),
                    'all': [round(t, 3) for t in t_port], 'code': 'oracle/restated.py'},
           'port_over_reference': round(med(t_port) / med(t_ref), 3),
           'update': f'{n_steps} optimisation steps per update of 131072 samples (the once-per-epoch tail, < 2 %, left out)'}
    with open(a.out, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
