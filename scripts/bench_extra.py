"""Supplementary measurements on one MI355X (not the headline bench line): the other BASELINE configurations and the
rollout-time inference path.  One JSON line per measurement on stdout.

    python scripts/bench_extra.py [--updates 4]

  amp   : config 1's agent at full yaml width (amp_humanoid.yaml: [1024, 512] nets, no latents), 4096 x 32 batch
  hrl   : config 4's high-level PPO update (obs 258, action 64, [1024, 512]), 4096 x 32 batch
  ase16k: config 5's batch (16384 envs x 32 = 524288 samples, 192 optimisation steps), ASE nets
  ase-f32 / ase-bf16x3: config 2 in the parity / split precision modes
  ase-mixed: config 2 under the reference's `mixed_precision: True` (f16 + dynamic loss scale)
  ase-dyn-gpx3: the headline mode (f16gpx3) with `loss_scale: dynamic` instead of its static gradient scale
  shard : ONE rank's share of the sharded data-parallel update (BASELINE configs[2]) on one GPU, for R = 2, 4, 8 ranks: the
          48 optimisation steps of an update at minibatch 16384 / R, amp minibatch 4096 / R (--shard-of R[,R...]); no collectives
          - the compute side of the scaling curve the driver's 8-GPU run would take, and the launch count it has to hide
  infer : get_action_values (eval-mode normalisation + actor + critic forward + sample) on 4096 observations
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ase_amd  # noqa: E402
ase_amd.configure(cpu_threads=1)     # hardware queues before HIP initialises; one torch CPU thread
import torch  # noqa: E402


def build(kind, num_envs, precision, graph=True, overrides=None, net_overrides=None):
    from ase_amd import cfg as defaults
    from ase_amd.learning import agents, models
    from ase_amd.learning.network_builder import AMPBuilder, ASEBuilder, HRLBuilder
    from ase_amd.synthetic import EnvSpec, SyntheticSource
    net_p, cfg = defaults.get(kind)
    cfg.update({k: v for k, v in (overrides or {}).items() if k == 'horizon_length'})
    for part, units in (net_overrides or {}).items():
        net_p[part]['units'] = list(units)
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
    cfg.update(overrides or {})
    info = {'observation_space': sp(obs), 'action_space': sp(act)}
    if amp:
        info['amp_observation_space'] = sp(amp)
    cfg.update(network=M(b), num_actors=num_envs, device='cuda:0', precision=precision, graph_capture=graph,
               vec_env=SyntheticSource(spec, seed=1236), env_info=info)
    if precision == 'mixed':          # the reference's own flag and nothing else: f16 storage + the dynamic loss scale (GradScaler)
        del cfg['precision']
        cfg['mixed_precision'] = True
    elif precision.endswith('+dyn'):  # a named mode with GradScaler's dynamic scale instead of its static one
        cfg['precision'] = precision[:-4]
        cfg['loss_scale'] = 'dynamic'
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
    ap.add_argument('--shard-of', default='', help='comma list of rank counts R: time one rank\'s share of the sharded update')
    ap.add_argument('--precision', default='bf16')
    ap.add_argument('--engine-opts', default='', help='JSON dict of UpdateEngine.engine_opts overrides (the --shard-of runs)')
    args = ap.parse_args()
    if args.shard_of:
        for R in [int(x) for x in args.shard_of.split(',')]:
            ov = {'minibatch_size': 16384 // R, 'amp_minibatch_size': 4096 // R}
            if args.engine_opts:
                ov['engine_opts'] = json.loads(args.engine_opts)
            ag, cfg, spec = build('ase', 4096, args.precision, overrides=ov)

            def one():
                return ag.update(ag._play_steps_tail(), max_steps=48)
            for _ in range(3):
                one()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.updates):
                one()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.updates
            progs = [g for g in ag._graphs.values() if not g['hipgraph']]
            entries = max((sum(ag.backend.prog_size(p) for p in g['graphs']) for g in progs), default=0)
            print(json.dumps({'measurement': f'shard-of-{R}', 'what': 'one rank\'s compute of the sharded data-parallel update: 48 '
                              f'optimisation steps at minibatch {16384 // R} / amp {4096 // R} rows + the epoch tail (whole batch: the tail '
                              'is sharded too in the real run), no collectives', 'ranks': R, 'ms_per_update': round(dt * 1e3, 3),
                              'us_per_step': round(dt * 1e6 / 48, 1), 'program_entries_per_step': entries, 'precision': args.precision,
                              'allreduce_bytes_per_step': int(ag.model.a2c_network.trainable_numel) * 4, 'gp_stream': bool(ag.engine._gp_side),
                              'ideal_us_per_step_from_1gpu': None}), flush=True)
            del ag
            torch.cuda.empty_cache()
        return
    # BASELINE configs[0] at its OWN size (64 envs x horizon 16, [256, 128] MLPs, minibatch 256 / amp 64, 24 optimisation steps):
    # the case the reference's CPU path is timed on (oracle/time_reference.py --config amp_cfg1) - launch-bound on a GPU
    cfg1 = dict(horizon_length=16, minibatch_size=256, mini_epochs=6, amp_minibatch_size=64, amp_batch_size=128,
                amp_obs_demo_buffer_size=512, amp_replay_buffer_size=2048)
    runs = [('amp', 'amp', 4096, 'bf16'), ('hrl', 'hrl', 4096, 'bf16'), ('ase16k', 'ase', 16384, 'bf16'),
            ('ase-f32', 'ase', 4096, 'f32'), ('ase-bf16x3', 'ase', 4096, 'bf16x3'), ('amp-cfg1', 'amp', 64, 'f32'),
            ('amp-cfg1-bf16', 'amp', 64, 'bf16'), ('ase-mixed', 'ase', 4096, 'mixed'), ('ase-dyn-gpx3', 'ase', 4096, 'f16gpx3+dyn')]
    for name, kind, envs, prec in runs:
        if args.only and name not in args.only.split(','):
            continue
        if name.startswith('amp-cfg1'):
            ag, cfg, spec = build(kind, envs, prec, overrides=cfg1, net_overrides={'mlp': [256, 128], 'disc': [256, 128]})
        else:
            ag, cfg, spec = build(kind, envs, prec, overrides=({'engine_opts': json.loads(args.engine_opts)} if args.engine_opts else None))
        dt = time_updates(ag, args.updates if (prec in ('bf16', 'mixed') or prec.endswith('+dyn')) else 2)
        B = ag.batch_size
        steps = cfg['mini_epochs'] * (B // cfg['minibatch_size'])
        print(json.dumps({'measurement': name, 'metric': 'PPO-update samples/sec', 'value': round(B / dt, 1), 'unit': 'samples/s',
                          'ms_per_update': round(dt * 1e3, 3), 'envs': envs, 'horizon': cfg['horizon_length'], 'batch': B,
                          'optimisation_steps': steps, 'precision': prec, 'params': int(ag.model.a2c_network.trainable_numel),
                          'replay': 'program', 'data': 'synthetic',
                          **({'loss_scaler': ag.engine.scaler_state(), 'what': ('config 2 under `mixed_precision: True`: f16 storage' if prec == 'mixed'
                              else 'config 2 in the headline mode with `loss_scale: dynamic`') + '; GradScaler on the device, per step: overflow '
                              'reported by the producing launches (scale records), skipped step + backoff / growth in front of the fused '
                              'optimizer launch, cross-step schedule kept'}
                             if ag.engine.dyn_scale else {})}), flush=True)
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
    if not args.only or 'ampobs' in args.only.split(','):
        # N2: observation production for 4096 envs (one frame + history push) and 5120 demo samples (512 x 10 steps)
        from ase_amd.backend import HipBackend
        be = HipBackend('cuda:0')
        g = torch.Generator().manual_seed(0)
        offs = [0, 3, 6, 9, 10, 13, 16, 17, 20, 21, 24, 27, 28, 31]
        N, D, K, S = 4096, 31, 6, 10
        q = torch.randn(N, 4, generator=g)
        q = (q / q.norm(dim=-1, keepdim=True)).cuda()
        st = [torch.randn(N, 3, generator=g).cuda(), q, torch.randn(N, 3, generator=g).cuda(), torch.randn(N, 3, generator=g).cuda(),
              (torch.randn(N, D, generator=g) * 0.5).cuda(), torch.randn(N, D, generator=g).cuda(), torch.randn(N, K, 3, generator=g).cuda()]
        hist = torch.zeros(N, S, 140, device='cuda')

        def timeit(fn, n=200):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n * 1e3
        us = timeit(lambda: be.build_amp_obs(*st, offs, True, True, hist, shift=True))
        byt = N * (S * 140 * 4 * 2 + (13 + 2 * D + 3 * K) * 4)
        print(json.dumps({'measurement': 'amp-obs frame + history push', 'envs': N, 'us_per_call': round(us, 1),
                          'GB_per_s': round(byt / us / 1e3, 1), 'bytes': byt}), flush=True)
        F_, B = 4000, 17
        lr = torch.randn(F_, B, 4, generator=g)
        clips = {'gts': torch.randn(F_, B, 3, generator=g).cuda(), 'grs': (lr / lr.norm(dim=-1, keepdim=True)).cuda(),
                 'lrs': (lr / lr.norm(dim=-1, keepdim=True)).cuda(), 'grvs': torch.randn(F_, 3, generator=g).cuda(),
                 'gravs': torch.randn(F_, 3, generator=g).cuda(), 'dvs': torch.randn(F_, D, generator=g).cuda(),
                 'lengths': torch.full((40,), 3.3).cuda(), 'dt': torch.full((40,), 1 / 30).cuda(),
                 'num_frames': torch.full((40,), 100, dtype=torch.int32).cuda(),
                 'length_starts': (torch.arange(40, dtype=torch.int32) * 100).cuda(),
                 'dof_body_ids': [1, 2, 3, 4, 5, 7, 8, 11, 12, 13, 14, 15, 16], 'dof_offsets': offs, 'key_body_ids': [5, 10, 13, 16, 6, 9]}
        n = 5120
        ids = torch.randint(0, 40, (n,), generator=g).to(torch.int32).cuda()
        tt = (torch.rand(n, generator=g) * 3.3).cuda()
        us = timeit(lambda: be.motion_state(clips, ids, tt))
        print(json.dumps({'measurement': 'motion-clip sampler', 'samples': n, 'us_per_call': round(us, 1)}), flush=True)
        # the wired pipeline: HumanoidAMP.fetch_amp_obs_demo for one demo-ring refill of config 2 (amp_batch_size 512 samples x 10 frames)
        from ase_amd.motion_lib import AmpObsDemoSource, DeviceMotionLib
        host = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in clips.items()}
        ml = DeviceMotionLib.from_arrays(host, be, 'cuda:0')
        src = AmpObsDemoSource(ml, be, num_amp_obs_steps=10, dt=1.0 / 30.0)
        for ns in (512, 4096):
            us = timeit(lambda: src.fetch_amp_obs_demo(ns), n=100)
            print(json.dumps({'measurement': 'fetch_amp_obs_demo (sample motions + times, motion state, 10-frame observations)',
                              'samples': ns, 'us_per_call': round(us, 1), 'out_bytes': ns * 1400 * 4}), flush=True)


if __name__ == '__main__':
    main()
