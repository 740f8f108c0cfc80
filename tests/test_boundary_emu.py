"""The plugin boundary beyond train_epoch, on CPU with the op emulator: the training loop rl_games'
Runner.run drives (``train()``), the rollout loop with the device-side action head (eps-greedy substitution, neglogp,
per-environment latents), players restored from agent checkpoints, and the HRL agent's ``env_step`` over a frozen low-level
controller - checked against plain-torch restatements of the reference statements they replace (cited inline)."""
import copy
import math
import os
import types

import pytest
import torch

from ase_amd.learning import agents, models, players
from ase_amd.synthetic import EnvSpec, SyntheticVecEnv
from tests.emu_backend import EmuBackend
from tests.helpers import BUILDERS, close

MODELS = {'ase': models.ModelASEContinuous, 'amp': models.ModelAMPContinuous, 'ppo': models.ModelHRLContinuous}
AGENTS = {'ase': agents.ASEAgent, 'amp': agents.AMPAgent, 'ppo': agents.CommonAgent}
PLAYERS = {'ase': players.ASEPlayer, 'amp': players.AMPPlayerContinuous, 'ppo': players.CommonPlayer}
GOLD = {'ase': 'ase_tiny', 'amp': 'amp_tiny', 'ppo': 'ppo_tiny'}
_DEV = 'cpu'                 # tests/test_gpu_boundary.py re-runs these functions with the HIP backend on cuda:0
_BE = EmuBackend


def _load(golden_dir, kind):
    return torch.load(os.path.join(golden_dir, GOLD[kind] + '.pt'), weights_only=False)


def _env(G, seed=3, task_obs=0):
    s = G['spec']
    spec = EnvSpec(num_envs=s['num_envs'], horizon=G['cfg']['horizon_length'], obs_size=s['obs_size'] - task_obs,
                   act_size=s['act_size'], amp_obs_size=s.get('amp_obs_size', 0) or 0,
                   latent_dim=G['cfg'].get('latent_dim', 0), latent_steps_min=G['cfg'].get('latent_steps_min', 1),
                   latent_steps_max=G['cfg'].get('latent_steps_max', 4), episode_length=5)
    return SyntheticVecEnv(spec, seed=seed, task_obs_size=task_obs, device=_DEV)


def _agent(G, env, be=None, **extra):
    kind = G['kind']
    b = BUILDERS[kind]()
    b.load(G['net'])
    cfg = dict(G['cfg'])
    info = {'observation_space': env.observation_space, 'action_space': env.action_space}
    if env.amp_observation_space is not None:
        info['amp_observation_space'] = env.amp_observation_space
    cfg.update(network=MODELS[kind](b), num_actors=env.num_envs, device=_DEV, backend=be or _BE(), precision='f32',
               env_info=info, vec_env=env, print_stats=False, seed=5)
    cfg.update(extra)
    return AGENTS[kind]('t', cfg), cfg


@pytest.mark.parametrize('kind', ['ase', 'amp', 'ppo'])
def test_train_loop_runs_like_runner(kind, golden_dir, tmp_path):
    """agent.train() (learning/common_agent.py:82-155): epochs of rollout + update until max_epochs, frame counters, reward
    meters, periodic checkpoint in the reference's dictionary layout."""
    G = _load(golden_dir, kind)
    env = _env(G)
    ag, cfg = _agent(G, env, max_epochs=2, save_frequency=1, train_dir=str(tmp_path), name='run')
    w0 = ag.model.a2c_network.flat_params.clone()
    last_mean_rewards, epoch_num = ag.train()
    assert epoch_num == 3 and ag.epoch_num == 3                       # the loop stops once epoch_num > max_epochs
    assert ag.frame == 3 * ag.batch_size
    assert env.steps == 3 * ag.horizon_length
    assert not torch.equal(w0, ag.model.a2c_network.flat_params)      # the optimizer ran
    assert ag.game_rewards.current_size > 0 and ag.game_lengths.get_mean() > 0
    for tag in ('losses/a_loss', 'losses/c_loss', 'info/kl', 'performance/total_fps', 'rewards0/frame'):
        assert tag in ag.writer.scalars, tag
    if kind != 'ppo':
        assert 'losses/disc_loss' in ag.writer.scalars and 'info/disc_reward_mean' in ag.writer.scalars
    ck = torch.load(os.path.join(str(tmp_path), 'run.pth'), weights_only=False)
    assert {'model', 'optimizer', 'epoch', 'frame', 'running_mean_std'} <= set(ck)
    assert ck['epoch'] == 3 and math.isfinite(float(ag.writer.scalars['losses/a_loss'][0]))


def test_rollout_action_head_semantics(golden_dir):
    """get_action_values(obs, ase_latents, rand_action_probs) (learning/ase_agent.py:117-148): deterministic rows take mu,
    the stored neglogp is the sampled action's, the mask follows the per-env probabilities, values are un-normalised."""
    G = _load(golden_dir, 'ase')
    env = _env(G)
    ag, cfg = _agent(G, env)
    ag.obs = ag.env_reset()
    n, A = env.num_envs, env.action_space.shape[0]
    z = ag._ase_latents
    assert torch.allclose(z.norm(dim=-1).cpu(), torch.ones(n), atol=1e-5)   # env_reset drew unit latents (ase_agent.py:310-321,352-360)
    probs = torch.zeros(n, device=_DEV)
    probs[: n // 2] = 1.0
    res = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in ag.get_action_values(ag.obs, z, probs).items()}
    mask = res['rand_action_mask']
    assert torch.equal(mask, probs)                                    # p = 1 -> stochastic, p = 0 -> deterministic
    det = mask == 0
    assert torch.equal(res['actions'][det], res['mus'][det])
    assert not torch.equal(res['actions'][~det], res['mus'][~det])
    # reference arithmetic on the same network outputs (rl_games neglogp, learning/amp_models.py:29-36)
    mu, sigma = res['mus'], res['sigmas']
    assert torch.allclose(sigma, torch.exp(ag.engine.logstd).expand_as(sigma))
    a = res['actions'][~det]
    nlp = 0.5 * (((a - mu[~det]) / sigma[~det]) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * A + torch.log(sigma[~det]).sum(-1)
    assert torch.allclose(res['neglogpacs'][~det], nlp, rtol=1e-5, atol=1e-5)
    # the same forward through the network-level API on pre-normalised observations (what the reference's model() sees)
    proc = ag._preproc_obs(ag.obs['obs'])
    mu_ref, _ = ag.model.a2c_network.eval_actor(proc, z)
    assert torch.allclose(mu, mu_ref, rtol=1e-5, atol=1e-6)
    v_ref = ag._eval_critic(ag.obs, z)
    assert torch.allclose(res['values'], v_ref, rtol=1e-5, atol=1e-6)
    # two calls draw different noise (the stream position advances on the device)
    res2 = ag.get_action_values(ag.obs, z, probs)
    assert not torch.equal(res2['actions'][~det], res['actions'][~det])


def test_latents_follow_progress(golden_dir):
    """_update_latents (learning/ase_agent.py:366-379): environments whose progress reached their reset step get a new
    latent and a later reset step; the others keep theirs."""
    G = _load(golden_dir, 'ase')
    env = _env(G)
    ag, cfg = _agent(G, env)
    ag.obs = ag.env_reset()
    z0, steps0 = ag._ase_latents.clone(), ag._latent_reset_steps.clone()
    env.progress_buf[:] = 0
    ag._update_latents()
    assert torch.equal(ag._ase_latents, z0)                            # reset steps are >= 1 > progress 0
    env.progress_buf[:4] = 100
    ag._update_latents()
    assert not torch.equal(ag._ase_latents[:4], z0[:4]) and torch.equal(ag._ase_latents[4:], z0[4:])
    assert bool((ag._latent_reset_steps[:4] > steps0[:4]).all()) and torch.equal(ag._latent_reset_steps[4:], steps0[4:])


@pytest.mark.parametrize('kind', ['ase', 'amp', 'ppo'])
def test_player_restores_agent_checkpoint(kind, golden_dir, tmp_path):
    """Players (learning/common_player.py, amp_players.py, ase_players.py) restore what the agent saved - weights and the
    input normalisers - and act with the same policy; run() plays episodes on the environment."""
    G = _load(golden_dir, kind)
    env = _env(G)
    ag, cfg = _agent(G, env, max_epochs=1, train_dir=str(tmp_path), name='p')
    ag.train()
    fn = os.path.join(str(tmp_path), 'p.pth')
    pcfg = dict(cfg)
    penv = _env(G, seed=9)
    pcfg.update(vec_env=penv, env_info=None, backend=_BE(), player={'games_num': 3, 'print_stats': False})
    if kind == 'ase':
        pcfg['env_info'] = None
    pl = PLAYERS[kind](pcfg)
    pl.restore(fn)
    assert torch.equal(pl.model.a2c_network.flat_params, ag.model.a2c_network.flat_params)
    assert torch.equal(pl.engine.obs_state, ag.engine.obs_state)
    if kind != 'ppo':
        assert torch.equal(pl.engine.amp_state, ag.engine.amp_state)
    obs = penv.reset()
    z = None
    if kind == 'ase':
        pl._reset_latents()
        z = pl._ase_latents
    act = pl.get_action({'obs': obs}, True) if kind != 'ase' else None
    mu = ag.engine.policy_forward(obs, z)['mu'].clone()
    if kind == 'ase':
        pl._latent_step_count = 5                                      # keep the latents through get_action
        act = pl.get_action({'obs': obs}, True)
    assert torch.allclose(act, torch.clamp(mu, -1.0, 1.0), rtol=1e-5, atol=1e-6)
    if kind != 'ppo':
        amp = penv.fetch_amp_obs_demo(6)
        assert torch.allclose(pl._calc_disc_rewards(amp), ag._calc_disc_rewards(amp), rtol=1e-5, atol=1e-6)
    pl.run()
    assert pl.games_played >= 3 and pl.sum_steps > 0


def _hrl_setup(golden_dir, tmp_path):
    GL = _load(golden_dir, 'ase')            # the low-level controller: the tiny ASE net
    GH = copy.deepcopy(_load(golden_dir, 'ppo'))
    task = 5
    sl = GL['spec']
    spec = EnvSpec(num_envs=GH['spec']['num_envs'], horizon=GH['cfg']['horizon_length'], obs_size=sl['obs_size'],
                   act_size=sl['act_size'], amp_obs_size=sl['amp_obs_size'], latent_dim=GL['cfg']['latent_dim'],
                   episode_length=7)
    env = SyntheticVecEnv(spec, seed=11, task_obs_size=task, device=_DEV)
    # an LLC checkpoint as ASEAgent.save writes it
    lenv = _env(GL, seed=2)
    llc, _ = _agent(GL, lenv, max_epochs=1, train_dir=str(tmp_path), name='llc')
    llc.train()
    llc_cfg = dict(GL['cfg'])
    llc_cfg['minibatch_size'] = spec.num_envs * GL['cfg']['horizon_length'] // 2
    llc_cfg['amp_minibatch_size'] = min(llc_cfg['amp_minibatch_size'], llc_cfg['minibatch_size'])
    b = BUILDERS['ppo']()
    b.load(GH['net'])
    cfg = dict(GH['cfg'])
    cfg.update(network=models.ModelHRLContinuous(b), num_actors=spec.num_envs, device=_DEV, backend=_BE(),
               precision='f32', vec_env=env, print_stats=False, seed=1,
               env_info={'observation_space': env.observation_space, 'action_space': env.action_space},
               llc_config={'params': {'network': GL['net'], 'config': llc_cfg}},
               llc_checkpoint=os.path.join(str(tmp_path), 'llc.pth'), llc_steps=3, task_reward_w=0.5, disc_reward_w=0.5,
               minibatch_size=spec.num_envs * GH['cfg']['horizon_length'] // 2)
    return cfg, env, llc, GL, GH


def test_hrl_env_step_with_frozen_llc(golden_dir, tmp_path):
    """HRLAgent.env_step / _compute_llc_action / _calc_disc_reward (learning/hrl_agent.py:45-82,231-249) against the same
    statements written with plain torch on the LLC's network-level API."""
    cfg, env, llc, GL, GH = _hrl_setup(golden_dir, tmp_path)
    ag = agents.HRLAgent('hrl', cfg)
    assert ag.actions_num == GL['cfg']['latent_dim'] and ag._task_size == 5
    L = ag._llc_agent
    assert torch.equal(L.model.a2c_network.flat_params, llc.model.a2c_network.flat_params)
    ag.obs = ag.env_reset()
    obs0 = ag.obs['obs'].clone()
    actions = (torch.randn(env.num_envs, ag.actions_num, generator=torch.Generator().manual_seed(0)) * 2.0).to(_DEV)
    # --- reference statements, step by step, on a twin environment
    twin = SyntheticVecEnv(env.spec, seed=11, task_obs_size=5, device=_DEV)
    twin.reset()
    a = torch.clamp(actions, -1.0, 1.0)
    obs = obs0
    rew = disc = dcount = tcount = 0.0
    for t in range(3):
        llc_obs = obs[..., :obs.shape[-1] - 5]
        proc = L._preproc_obs(llc_obs)
        z = torch.nn.functional.normalize(a, dim=-1)
        mu, _ = L.model.a2c_network.eval_actor(obs=proc, ase_latents=z)
        llc_action = agents.rescale_actions(L.actions_low, L.actions_high, torch.clamp(mu, -1.0, 1.0))
        o, r, d, info = twin.step(llc_action)
        obs = o
        rew = rew + r
        dcount = dcount + d.float()
        tcount = tcount + info['terminate'].float()
        disc = disc + L._calc_disc_rewards(info['amp_obs'])
    # --- the agent
    new_obs, rewards, dones, infos = ag.env_step(actions)
    assert torch.allclose(env.last_actions, twin.last_actions, rtol=1e-5, atol=1e-6)       # same LLC actions reached the env
    assert torch.equal(new_obs['obs'], obs)
    assert torch.allclose(rewards.view(-1), rew / 3) and rewards.shape == (env.num_envs, 1)
    assert torch.equal(dones, (dcount > 0).float()) and torch.equal(infos['terminate'], (tcount > 0).float())
    assert torch.allclose(infos['disc_rewards'], disc / 3, rtol=1e-5, atol=1e-6)


def test_hrl_train_and_player(golden_dir, tmp_path):
    """Config 4 end to end on the synthetic environment: HRL rollout (5 -> 3 inner LLC steps per high-level step), reward mix
    task_reward_w * r + disc_reward_w * r_disc into GAE, plain PPO update of the high-level net; HRLPlayer on the result."""
    cfg, env, llc, GL, GH = _hrl_setup(golden_dir, tmp_path)
    cfg.update(max_epochs=1, train_dir=str(tmp_path), name='hlc')
    ag = agents.HRLAgent('hrl', cfg)
    w_llc = ag._llc_agent.model.a2c_network.flat_params.clone()
    ag.train()
    assert env.steps == 2 * ag.horizon_length * 3
    assert torch.equal(w_llc, ag._llc_agent.model.a2c_network.flat_params)                  # the LLC stayed frozen
    E = ag.experience
    exp_r = 0.5 * E['rewards'] + 0.5 * E['disc_rewards']
    # GAE with the mixed reward, last step of the horizon (learning/common_agent.py:437-449): adv_T = r_T + gamma next_v_T - v_T
    T = ag.horizon_length - 1
    adv_T = exp_r[T] + ag.gamma * E['next_values'][T] - E['values'][T]
    assert torch.allclose(ag._tail_info['mb_advs'].view(ag.horizon_length, -1, 1)[T], adv_T, rtol=1e-4, atol=1e-5)
    pcfg = dict(cfg)
    pcfg.update(vec_env=SyntheticVecEnv(env.spec, seed=4, task_obs_size=5, device=_DEV), env_info=None, backend=_BE(),
                player={'games_num': 2, 'print_stats': False})
    pl = players.HRLPlayer(pcfg)
    pl.restore(os.path.join(str(tmp_path), 'hlc.pth'))
    assert torch.equal(pl.model.a2c_network.flat_params, ag.model.a2c_network.flat_params)
    pl.run()
    assert pl.games_played >= 2


def test_hrl_env_step_matches_reference_golden(golden_dir):
    """tests/golden/hrl_step.pt was recorded from the REFERENCE'S OWN HRLAgent.env_step / _compute_llc_action /
    _calc_disc_reward (learning/hrl_agent.py:45-82,231-249, oracle/make_golden_hrl.py) over a frozen reference ASE agent on
    the seeded synthetic environment: same LLC weights and running statistics, same environment seed, same high-level
    actions -> the same low-level actions reach the environment, the same rewards / dones / terminate / discriminator
    rewards come back."""
    G = torch.load(os.path.join(golden_dir, 'hrl_step.pt'), weights_only=False)
    sp, L, Hc = G['spec'], G['llc'], G['hlc']
    spec = EnvSpec(num_envs=sp['num_envs'], horizon=Hc['cfg']['horizon_length'], obs_size=sp['obs_size'], act_size=sp['act_size'],
                   amp_obs_size=sp['amp_obs_size'], latent_dim=L['cfg']['latent_dim'], episode_length=G['episode_length'])
    env = SyntheticVecEnv(spec, seed=G['env_seed'], task_obs_size=sp['task'], device=_DEV)
    llc_ckpt = {'model': L['sd'], 'running_mean_std': L['running_mean_std'], 'amp_input_mean_std': L['amp_input_mean_std'],
                'reward_mean_std': L['reward_mean_std'], 'epoch': 0, 'frame': 0,
                'optimizer': {'state': {}, 'param_groups': [{'lr': L['cfg']['learning_rate']}]}}
    b = BUILDERS['ppo']()
    b.load(Hc['net'])
    cfg = dict(Hc['cfg'])
    cfg.update(network=models.ModelHRLContinuous(b), num_actors=spec.num_envs, device=_DEV, backend=_BE(), precision='f32',
               vec_env=env, print_stats=False, env_info={'observation_space': env.observation_space, 'action_space': env.action_space},
               llc_config={'params': {'network': L['net'], 'config': dict(L['cfg'])}}, llc_checkpoint=llc_ckpt,
               llc_steps=G['llc_steps'])
    ag = agents.HRLAgent('hrl', cfg)
    ag.obs = ag.env_reset()
    assert torch.equal(ag.obs['obs'].cpu(), G['obs0'])
    actions = G['actions'].to(_DEV)
    a0 = ag._compute_llc_action(ag.obs['obs'], ag.preprocess_actions(actions))
    assert torch.allclose(a0.cpu(), G['llc_action0'], rtol=1e-4, atol=2e-5)
    obs1, rewards, dones, infos = ag.env_step(actions)
    assert torch.allclose(env.last_actions.cpu(), G['env_last_actions'], rtol=1e-4, atol=2e-5)
    assert torch.equal(obs1['obs'].cpu(), G['obs1'])
    assert torch.allclose(rewards.cpu(), G['rewards']) and torch.equal(dones.cpu().float(), G['dones'].float())
    assert torch.equal(infos['terminate'].cpu().float(), G['terminate'].float())
    assert torch.allclose(infos['disc_rewards'].cpu(), G['disc_rewards'], rtol=1e-4, atol=2e-5)


# ------------------------------------------------------------------------------------------------ N2: demo AMP observations
def _demo_source(be, dev, local_root, root_h, golden_dir, generator=None):
    from ase_amd.motion_lib import AmpObsDemoSource, DeviceMotionLib
    M = torch.load(os.path.join(golden_dir, 'motion_state.pt'), weights_only=False)
    G = torch.load(os.path.join(golden_dir, 'amp_obs_demo.pt'), weights_only=False)
    ml = DeviceMotionLib.from_arrays(M['clips'], be, dev, weights=G['weights'], generator=generator)
    src = AmpObsDemoSource(ml, be, num_amp_obs_steps=G['steps'], dt=G['dt'], local_root_obs=local_root, root_height_obs=root_h)
    return ml, src, G


@pytest.mark.parametrize('local_root,root_h', [(True, True), (False, False)])
def test_fetch_amp_obs_demo_matches_reference(local_root, root_h, golden_dir):
    """DeviceMotionLib.sample_motions / sample_time + AmpObsDemoSource.fetch_amp_obs_demo against the
    reference's HumanoidAMP.fetch_amp_obs_demo (humanoid_amp.py:63-105; golden = the reference's own methods over its own
    MotionLib on two shipped clips).  On the host generator (CPU run) the sampler draws the same motions and times as the
    reference under the same torch seed and the whole fetch is compared; on the GPU the draws come from the device generator,
    so the observations are compared at the golden's (motion, time) pairs and the fetch is checked for shape and range."""
    ml, src, G = _demo_source(_BE(), _DEV, local_root, root_h, golden_dir)
    case = G['cases'][(local_root, root_h)]
    n, S, dt = G['n'], G['steps'], G['dt']
    assert ml.num_motions() == 2 and src.get_num_amp_obs() == 1400
    ids_g, t0_g = case['motion_ids'].to(_DEV), case['motion_times0'].to(_DEV)
    want = case['amp_obs_demo']
    if _DEV == 'cpu':
        torch.manual_seed(G['seed'])
        ids = ml.sample_motions(n)
        t0 = ml.sample_time(ids, truncate_time=dt * (S - 1)) + dt * (S - 1)
        assert torch.equal(ids, case['motion_ids'])
        assert torch.allclose(t0, case['motion_times0'], rtol=0, atol=1e-6)
        torch.manual_seed(G['seed'])
        out = src.fetch_amp_obs_demo(n)
        assert torch.allclose(out, want, rtol=1e-5, atol=1e-5)
    # the observations at the reference's samples (interpolated rotations go through acos / sin on both sides: 1e-4 on device)
    obs = src.build_amp_obs_demo(ids_g, t0_g)
    assert obs.shape == (n, S, 140)
    assert torch.allclose(obs.reshape(n, -1).cpu(), want, rtol=1e-4, atol=1e-4)
    # newest frame first: slot k of a sample is the state k * dt before the sampled time
    first = obs[:3, 1].clone()
    late = src.build_amp_obs_demo(ids_g[:3], t0_g[:3] - dt)
    assert torch.allclose(late[:, 0], first, atol=1e-4)
    # a fresh fetch: right shape, finite, times inside [truncate, clip length], both clips drawn with the stated weights
    big = 4096
    ids = ml.sample_motions(big)
    t0 = ml.sample_time(ids, truncate_time=dt * (S - 1)) + dt * (S - 1)
    lens = ml.get_motion_length(ids)
    assert bool((t0 >= dt * (S - 1) - 1e-6).all()) and bool((t0 <= lens + 1e-6).all())
    frac1 = float((ids == 1).float().mean())
    assert abs(frac1 - float(G['weights'][1])) < 0.05
    out = src.fetch_amp_obs_demo(big)
    assert out.shape == (big, 1400) and bool(torch.isfinite(out).all())
    # the env hook the agents call (amp_agent.py:498-500 -> vec_env.env.fetch_amp_obs_demo)
    from ase_amd.synthetic import EnvSpec, SyntheticVecEnv
    env = SyntheticVecEnv(EnvSpec(num_envs=8, horizon=2), device=_DEV, demo_source=src)
    assert env.fetch_amp_obs_demo(16).shape == (16, 1400)


def test_motion_lib_from_reference_object(golden_dir):
    """DeviceMotionLib.from_reference takes a loaded reference MotionLib (duck-typed here: the attributes its loader
    leaves, utils/motion_lib.py:174-252) - same state as from the arrays, weights normalised like motion_lib.py:213."""
    import types
    from ase_amd.motion_lib import DeviceMotionLib
    M = torch.load(os.path.join(golden_dir, 'motion_state.pt'), weights_only=False)
    c = M['clips']
    ref = types.SimpleNamespace(gts=c['gts'], grs=c['grs'], lrs=c['lrs'], grvs=c['grvs'], gravs=c['gravs'], dvs=c['dvs'],
                                _motion_lengths=c['lengths'], _motion_num_frames=c['num_frames'], _motion_dt=c['dt'],
                                length_starts=c['length_starts'], _motion_weights=torch.tensor([1.0, 3.0]))
    be = _BE()
    a = DeviceMotionLib.from_reference(ref, be, _DEV, c['dof_body_ids'], c['dof_offsets'], c['key_body_ids'])
    b = DeviceMotionLib.from_arrays(c, be, _DEV, weights=[0.25, 0.75])
    assert torch.allclose(a._motion_weights.cpu(), torch.tensor([0.25, 0.75])) and a.num_motions() == b.num_motions() == 2
    assert abs(a.get_total_length() - float(c['lengths'].sum())) < 1e-5
    ids, t = M['motion_ids'][:16].to(_DEV), M['times'][:16].to(_DEV)
    for x, y in zip(a.get_motion_state(ids, t), b.get_motion_state(ids, t)):
        assert torch.equal(x, y)


def test_precision_resolver_is_shared_by_trainer_and_player():
    """One resolver (ase_amd/cfg): f16gp32 / f16gpx3 play in f16, `mixed_precision: True` without a precision key means f16 in
    both the agent and the player, unknown names raise instead of silently running another arithmetic."""
    from ase_amd.cfg import resolve_precision
    assert resolve_precision({}) == ('bf16', torch.bfloat16)
    assert resolve_precision({'mixed_precision': True}) == ('f16', torch.float16)
    assert resolve_precision({'mixed_precision': True, 'precision': 'f32'}) == ('f32', torch.float32)
    for p in ('f16gp32', 'f16gpx3'):
        assert resolve_precision({'precision': p}) == (p, torch.float16)
    assert resolve_precision({'precision': 'bf16x3'})[1] == torch.float32
    with pytest.raises(ValueError):
        resolve_precision({'precision': 'fp8'})
    import inspect
    from ase_amd.learning import agents, players
    assert 'resolve_precision' in inspect.getsource(agents.CommonAgent.__init__)
    assert 'resolve_precision' in inspect.getsource(players)


@pytest.mark.parametrize('name', ['ase_tiny', 'amp_tiny', 'ppo_tiny'])
def test_model_wrapper_forward_matches_the_reference(name, golden_dir):
    """learning/models.py `Network.forward` against the REFERENCE'S OWN model wrappers (learning/ase_models.py:19-29,
    amp_models.py:20-38, hrl_models.py; tests/golden/model_forward.pt from oracle/make_golden_model.py): same result keys, train
    mode - prev_neglogp, values, entropy, mus, sigmas, the three discriminator logits, enc_pred - and play mode - mus / sigmas /
    values, and the returned neglogpacs are those of the returned actions."""
    import math
    from tests.test_agent_emu import make_agent
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    M = torch.load(os.path.join(golden_dir, 'model_forward.pt'), weights_only=False)[name]
    ag = make_agent(G, EmuBackend(), precision='f32')
    model = ag.model
    out = model(dict(M['inputs'], is_train=True))
    assert set(out) == set(M['train']), (sorted(out), sorted(M['train']))
    for k, ref in M['train'].items():
        if ref is None:
            assert out[k] is None, k
            continue
        assert out[k].shape == ref.shape, (k, out[k].shape, ref.shape)
        close(out[k], ref, 2e-5, 2e-5 * max(1.0, float(ref.abs().max())), 'train ' + k)
    torch.manual_seed(M['play_seed'])
    play = model(dict(M['inputs'], is_train=False))
    assert set(play) == set(M['play'])
    for k in ('mus', 'sigmas', 'values'):
        close(play[k], M['play'][k], 2e-5, 2e-5 * max(1.0, float(M['play'][k].abs().max())), 'play ' + k)
    a, mu, sg = play['actions'], play['mus'], play['sigmas']
    nlp = 0.5 * (((a - mu) / sg) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * a.shape[-1] + sg.log().sum(-1)
    close(play['neglogpacs'], nlp, 1e-5, 1e-5, 'play neglogpacs of the returned actions')
    assert play['neglogpacs'].shape == M['play']['neglogpacs'].shape and a.shape == M['play']['actions'].shape
    # the draw itself: mu + sigma * N(0, 1) from torch's generator, like Normal(mu, sigma).sample() - same seed, same actions
    close(a, M['play']['actions'], 2e-5, 2e-5 * max(1.0, float(M['play']['actions'].abs().max())), 'play actions (seeded)')


@pytest.mark.parametrize('kind', ['ase', 'amp'])
def test_player_reward_helpers_match_the_oracle(kind, golden_dir, tmp_path):
    """The players' stand-alone discriminator / encoder helpers (learning/amp_players.py:60-110 `_preproc_amp_obs`, `_eval_disc`,
    `_calc_disc_rewards`, `_calc_amp_rewards`; learning/ase_players.py `_eval_enc`, `_calc_enc_rewards`) on a restored checkpoint
    against oracle/restated.py's `calc_disc_rewards` / `calc_enc_rewards` with the checkpoint's weights and AMP statistics."""
    from oracle import restated as R
    from tests.helpers import get_rms
    G = _load(golden_dir, kind)
    env = _env(G)
    ag, cfg = _agent(G, env, max_epochs=1, train_dir=str(tmp_path), name='q')
    ag.train()
    pcfg = dict(cfg)
    pcfg.update(vec_env=_env(G, seed=9), env_info=None, backend=_BE(), player={'games_num': 1, 'print_stats': False})
    pl = PLAYERS[kind](pcfg)
    pl.restore(os.path.join(str(tmp_path), 'q.pth'))
    sd = R.canonical_sd({k: v.detach().cpu() for k, v in pl.model.state_dict().items()}, False)
    rms = get_rms(pl.engine.amp_state)
    rms = {'mean': rms['mean'].float(), 'var': rms['var'].float(), 'count': rms['count']}
    amp = env.fetch_amp_obs_demo(12)
    amp_c = amp.cpu()
    # the eval-mode AMP normaliser on its own
    x = pl._preproc_amp_obs(amp)
    assert torch.allclose(x.cpu(), R.rms_normalize(rms, amp_c), rtol=1e-5, atol=1e-5)
    ref_d = R.calc_disc_rewards(sd, rms, amp_c, G['cfg']['disc_reward_scale'])
    out = pl._calc_amp_rewards(amp) if kind == 'amp' else None
    if kind == 'amp':
        assert set(out) == {'disc_rewards'}
        assert torch.allclose(out['disc_rewards'].cpu(), ref_d, rtol=1e-5, atol=1e-6)
        assert torch.allclose(pl._eval_disc(amp).cpu(), R.eval_disc(sd, R.rms_normalize(rms, amp_c)), rtol=1e-5, atol=1e-5)
        return
    g = torch.Generator().manual_seed(2)
    z = torch.nn.functional.normalize(torch.randn(12, G['cfg']['latent_dim'], generator=g), dim=-1)
    out = pl._calc_amp_rewards(amp, z.to(amp.device))
    assert set(out) == {'disc_rewards', 'enc_rewards'}
    assert torch.allclose(out['disc_rewards'].cpu(), ref_d, rtol=1e-5, atol=1e-6)
    assert torch.allclose(out['enc_rewards'].cpu(), R.calc_enc_rewards(sd, rms, amp_c, z, G['cfg']['enc_reward_scale']), rtol=1e-5, atol=1e-6)
    assert torch.allclose(pl._eval_enc(amp).cpu(), R.eval_enc(sd, R.rms_normalize(rms, amp_c)), rtol=1e-5, atol=1e-5)


def test_network_weight_getters_and_initialisers(golden_dir):
    """The network-level helpers the reference's losses call (learning/amp_network_builder.py:86-96 `get_disc_logit_weights`,
    `get_disc_weights`; learning/ase_network_builder.py `get_enc_weights`) return the same tensors in the same order, a FRESH
    network starts where the reference's initialisers put it (zero biases; logit / encoder / style-dense weights uniform within
    the reference's constants), and the flat parameter buffer refuses to move once an engine has bound it."""
    G = _load(golden_dir, 'ase')
    b = BUILDERS['ase']()
    b.load(G['net'])
    s = G['spec']
    torch.manual_seed(0)
    net = b.build('ase', actions_num=s['act_size'], input_shape=(s['obs_size'],), num_seqs=s['num_envs'], value_size=1,
                  amp_input_shape=(s['amp_obs_size'],), ase_latent_shape=(G['cfg']['latent_dim'],), device='cpu')
    sd = net.state_dict()
    disc_keys = sorted(k for k in sd if k.startswith('_disc_mlp.') and k.endswith('.weight'))
    ws = net.get_disc_weights()
    assert len(ws) == len(disc_keys) + 1
    for w, k in zip(ws, disc_keys + ['_disc_logits.weight']):
        assert torch.equal(w, sd[k].reshape(-1)), k
    assert torch.equal(net.get_disc_logit_weights(), sd['_disc_logits.weight'].reshape(-1))
    enc_keys = sorted(k for k in sd if k.startswith('_enc_mlp.') and k.endswith('.weight'))
    we = net.get_enc_weights()
    assert len(we) == len(enc_keys) + 1 and torch.equal(we[-1], sd['_enc.weight'].reshape(-1))
    for w, k in zip(we, enc_keys):
        assert torch.equal(w, sd[k].reshape(-1)), k
    # shared trunk: the encoder's trunk IS the discriminator's (checkpoint aliases, learning/ase_network_builder.py:202-203)
    assert all(torch.equal(sd[k], sd[k.replace('_enc_mlp.', '_disc_mlp.')]) for k in enc_keys)
    # initialisers
    from ase_amd.learning import network_builder as NB
    assert all(float(v.abs().max()) == 0.0 for k, v in sd.items() if k.endswith('.bias'))
    assert float(sd['_disc_logits.weight'].abs().max()) <= NB.DISC_LOGIT_INIT_SCALE
    assert float(sd['_enc.weight'].abs().max()) <= NB.ENC_LOGIT_INIT_SCALE < 0.5
    assert 0.5 < float(sd['actor_mlp._style_dense.weight'].abs().max()) <= NB.STYLE_INIT_RANGE
    assert abs(float(sd['sigma'][0]) - G['net']['space']['continuous']['sigma_init']['val']) < 1e-6 and not net.sigma.requires_grad
    if os.path.isdir('/root/reference/ase/learning'):            # the constants, read from the reference's own sources
        import re
        src = lambda f: open(os.path.join('/root/reference/ase/learning', f)).read()
        assert NB.DISC_LOGIT_INIT_SCALE == float(re.search(r'^DISC_LOGIT_INIT_SCALE\s*=\s*([0-9.]+)', src('amp_network_builder.py'), re.M).group(1))
        assert NB.ENC_LOGIT_INIT_SCALE == float(re.search(r'^ENC_LOGIT_INIT_SCALE\s*=\s*([0-9.]+)', src('ase_network_builder.py'), re.M).group(1))
        assert NB.STYLE_INIT_RANGE == float(re.search(r'scale_init_range\s*=\s*([0-9.]+)', src('ase_network_builder.py')).group(1))
    # an engine binds the flat buffer: .to() / .double() afterwards must fail loudly instead of orphaning its shadows
    ag, _ = _agent(G, _env(G))
    with pytest.raises((RuntimeError, TypeError)):
        ag.model.a2c_network.double()
