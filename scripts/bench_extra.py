"""Supplementary measurements on one MI355X (not the headline bench line): the other BASELINE configurations and the
rollout-time inference path.  One JSON line per measurement on stdout.

    python scripts/bench_extra.py [--updates 4]

  amp   : config 1's agent at full yaml width (amp_humanoid.yaml: [1024, 512] nets, no latents), 4096 x 32 batch
  hrl   : config 4's high-level PPO update (obs 258, action 64, [1024, 512]), 4096 x 32 batch
  ase16k: config 5's batch (16384 envs x 32 = 524288 samples, 192 optimisation steps), ASE nets
  ase-f32 / ase-bf16x3: config 2 in the parity / split precision modes
  infer : get_action_values (eval-mode normalisation + actor + critic forward + sample) on 4096 observations
"""
import argparse
import json
import os
import sys
import time
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(kind, num_envs, precision, graph=True):
    from ase_amd import cfg as defaults
    from ase_amd.learning import agents, models
    from ase_amd.learning.network_builder import AMPBuilder, ASEBuilder, HRLBuilder
    from ase_amd.synthetic import EnvSpec, SyntheticSource
    net_p, cfg = defaults.get(kind)
    obs, act, amp = {'ase': (253, 31, 1400), 'amp': (253, 31, 1400), 'hrl': (258, 64, 0)}[kind]
    z = cfg.get('latent_dim', 0) if kind == 'ase' else 0
    spec = EnvSpec(num_envs=num_envs, horizon=cfg['horizon_length'], obs_size=obs, act_size=act, amp_obs_size=amp,
                   latent_dim=z, latent_steps_min=cfg.get('latent_steps_min', 1), latent_steps_max=cfg.get('latent_steps_max', 150))
    torch.manual_seed(0)
    B, M, A = {'ase': (ASEBuilder, models.ModelASEContinuous, agents.ASEAgent),
               'amp': (AMPBuilder, models.ModelAMPContinuous, agents.AMPAgent),
               'hrl': (HRLBuilder, models.ModelHRLContinuous, agents.CommonAgent)}[kind]
    b = B()
    b.load(net_p)
    sp = lambda n: types.SimpleNamespace(shape=(n,))
    cfg = dict(cfg)
    info = {'observation_space': sp(obs), 'action_space': sp(act)}
    if amp:
        info['amp_observation_space'] = sp(amp)
    cfg.update(network=M(b), num_actors=num_envs, device='cuda:0', precision=precision, graph_capture=graph,
               vec_env=SyntheticSource(spec, seed=1236), env_info=info)
    ag = A('extra', cfg)
    with torch.no_grad():
        ag.set_eval()
        exp = ag.vec_env.experience(ag._cpu_policy(), **ag._experience_kwargs())
        for k, v in exp.items():
            if k in ag.experience:
                ag.experience[k].copy_(v.to('cuda:0'))
        if amp:
            ag._init_amp_demo_buf()
    torch.cuda.synchronize()
    return ag, cfg, spec


def time_updates(ag, n):
    def one():
        return ag.update(ag._play_steps_tail())
    for _ in range(4):          # graph capture (two replay-source variants) + warm-up
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        one()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--updates', type=int, default=4)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    runs = [('amp', 'amp', 4096, 'bf16'), ('hrl', 'hrl', 4096, 'bf16'), ('ase16k', 'ase', 16384, 'bf16'),
            ('ase-f32', 'ase', 4096, 'f32'), ('ase-bf16x3', 'ase', 4096, 'bf16x3')]
    for name, kind, envs, prec in runs:
        if args.only and name not in args.only.split(','):
            continue
        ag, cfg, spec = build(kind, envs, prec)
        dt = time_updates(ag, args.updates if prec == 'bf16' else 2)
        B = ag.batch_size
        steps = cfg['mini_epochs'] * (B // cfg['minibatch_size'])
        print(json.dumps({'measurement': name, 'metric': 'PPO-update samples/sec', 'value': round(B / dt, 1), 'unit': 'samples/s',
                          'ms_per_update': round(dt * 1e3, 3), 'envs': envs, 'horizon': cfg['horizon_length'], 'batch': B,
                          'optimisation_steps': steps, 'precision': prec, 'params': int(ag.model.a2c_network.trainable_numel),
                          'hipgraph': True, 'data': 'synthetic'}), flush=True)
        del ag
        torch.cuda.empty_cache()
    if not args.only or 'infer' in args.only.split(','):
        ag, cfg, spec = build('ase', 4096, 'bf16', graph=False)
        obs = ag.experience['obses'][0].contiguous()
        z = ag.experience['ase_latents'][0].contiguous()
        ag.set_eval()
        for _ in range(5):
            ag.get_action_values({'obs': obs}, z)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 200
        for _ in range(n):
            ag.get_action_values({'obs': obs}, z)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(json.dumps({'measurement': 'infer', 'metric': 'rollout inference (get_action_values) env-steps/sec',
                          'value': round(4096 / dt, 1), 'unit': 'env-steps/s', 'us_per_call': round(dt * 1e6, 1), 'envs': 4096,
                          'precision': 'bf16', 'hipgraph': False}), flush=True)


if __name__ == '__main__':
    main()
