"""Seeded synthetic rollout / demo generator (Isaac Gym stand-in).

BASELINE.json puts the simulator out of scope: the experience buffer that the reference fills in
``play_steps`` (learning/ase_agent.py:36-93 under /root/reference/ase) is produced here with the
same tensor names, shapes and dtypes (rl_games ExperienceBuffer layout, time-major ``[H, N, ...]``),
so the update path — everything after the rollout loop — sees exactly what it would see in
training.  Distributions follow SURVEY.md §8(d).

Everything is generated on CPU from one ``torch.Generator`` (identical bits for the CPU oracle and
the HIP path), then moved by the caller.  The policy outputs stored with the experience
(``values``, ``mus``, ``sigmas``, ``neglogpacs``, ``actions``) come from a caller-supplied
``policy`` so that the importance ratio is ~1 on the first optimisation step, as in real training.
"""
import math

import torch


class EnvSpec:
    """Shapes of one (task, character) pair.  Defaults = HumanoidAMPGetup sword&shield
    (ase/data/cfg/humanoid_ase_sword_shield_getup.yaml:3,14,20): obs 253, act 31, amp 10x140."""

    def __init__(self, num_envs=4096, horizon=32, obs_size=253, act_size=31, amp_obs_size=1400,
                 latent_dim=64, latent_steps_min=1, latent_steps_max=150, episode_length=300):
        self.num_envs, self.horizon = num_envs, horizon
        self.obs_size, self.act_size, self.amp_obs_size = obs_size, act_size, amp_obs_size
        self.latent_dim = latent_dim
        self.latent_steps_min, self.latent_steps_max = latent_steps_min, latent_steps_max
        self.episode_length = episode_length


def rand_action_probs(num_envs, enable_eps_greedy=True):
    """Per-env probability of taking the stochastic action (learning/amp_agent.py:424-435)."""
    env_ids = torch.arange(num_envs, dtype=torch.float32)
    p = 1.0 - torch.exp(10 * (env_ids / (num_envs - 1.0) - 1.0))
    p[0] = 1.0
    p[-1] = 0.0
    if not enable_eps_greedy:
        p[:] = 1.0
    return p


def _unit_rows(x):
    return x / x.norm(dim=-1, keepdim=True).clamp_min(1e-12)


class FeatureScale:
    """Per-feature scale s_j ~ U(0.5, 2) and offset o_j ~ U(-1, 1)."""

    def __init__(self, size, gen):
        self.scale = torch.rand(size, generator=gen) * 1.5 + 0.5
        self.offset = torch.rand(size, generator=gen) * 2.0 - 1.0

    def draw(self, n_rows, gen, mean=0.0):
        x = torch.randn(n_rows, self.scale.numel(), generator=gen)
        if mean != 0.0:
            x += mean
        return x * self.scale + self.offset


class SyntheticSource:
    """Holds the per-feature statistics so that agent / demo / later epochs share them."""

    def __init__(self, spec, seed=1234):
        self.spec = spec
        self.gen = torch.Generator().manual_seed(seed)
        self.obs_fs = FeatureScale(spec.obs_size, self.gen)
        self.amp_fs = FeatureScale(spec.amp_obs_size, self.gen) if spec.amp_obs_size else None
        self.probs = rand_action_probs(spec.num_envs)

    # -- demo stream: env.fetch_amp_obs_demo stand-in (learning/amp_agent.py:498-500) --
    def fetch_amp_obs_demo(self, n):
        # distinct mean so the discriminator has a signal
        return self.amp_fs.draw(n, self.gen, mean=0.3)

    def latents(self):
        """[H, N, z]: unit vectors held constant per env over spans ~ U{min..max-1}
        (learning/ase_agent.py:323-326,366-379)."""
        s = self.spec
        H, N, D = s.horizon, s.num_envs, s.latent_dim
        z = torch.empty(H, N, D)
        cur = _unit_rows(torch.randn(N, D, generator=self.gen))
        left = torch.randint(s.latent_steps_min, s.latent_steps_max, (N,), generator=self.gen)
        for t in range(H):
            renew = left <= 0
            k = int(renew.sum())
            if k:
                cur[renew] = _unit_rows(torch.randn(k, D, generator=self.gen))
                left[renew] = torch.randint(s.latent_steps_min, s.latent_steps_max, (k,), generator=self.gen)
            z[t] = cur
            left -= 1
        return z

    def experience(self, policy, with_amp=True, with_latents=True):
        """One rollout's worth of buffers.  ``policy(obs[M,obs], z[M,zdim] or None)`` must return
        (mu[M,act], sigma[M,act], value[M,1]) for *raw* observations — it applies whatever
        observation / value (un)normalisation the agent uses (learning/ase_agent.py:117-148)."""
        s = self.spec
        H, N = s.horizon, s.num_envs
        g = self.gen
        exp = {}
        obs = self.obs_fs.draw(H * N, g).view(H, N, -1)
        nxt = self.obs_fs.draw(H * N, g).view(H, N, -1)
        exp['obses'], exp['next_obses'] = obs, nxt
        z = self.latents() if with_latents else None
        zf = z.reshape(H * N, -1) if z is not None else None
        mu, sigma, value = policy(obs.reshape(H * N, -1), zf)
        _, _, nvalue = policy(nxt.reshape(H * N, -1), zf)
        noise = torch.randn(H * N, s.act_size, generator=g)
        actions = mu + sigma * noise
        logstd = torch.log(sigma)
        nlp = 0.5 * (((actions - mu) / sigma) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * s.act_size + logstd.sum(-1)
        mask = torch.bernoulli(self.probs.expand(H, N), generator=g)
        det = (mask == 0.0).reshape(H * N)
        if with_amp:                                   # eps-greedy belongs to the AMP / ASE agents (learning/ase_agent.py:143-146:
            actions[det] = mu[det]                     # the mean action, the SAMPLE's neglogp, and the row masked out of the
                                                       # actor loss); a plain PPO rollout always acts on its samples
        dones = (torch.rand(H, N, generator=g) < 1.0 / s.episode_length)
        terminate = dones & (torch.rand(H, N, generator=g) < 0.5)
        exp['actions'] = actions.view(H, N, -1)
        exp['neglogpacs'] = nlp.view(H, N)
        exp['values'] = value.view(H, N, 1)
        exp['mus'] = mu.reshape(H, N, -1).clone()
        exp['sigmas'] = sigma.reshape(H, N, -1).clone()
        exp['dones'] = dones.to(torch.uint8)
        exp['rewards'] = torch.ones(H, N, 1)            # humanoid.py:638-642: task reward == 1
        exp['next_values'] = nvalue.view(H, N, 1) * (1.0 - terminate.float().unsqueeze(-1))
        exp['rand_action_mask'] = mask
        if with_amp:
            exp['amp_obs'] = self.amp_fs.draw(H * N, g).view(H, N, -1)
        if z is not None:
            exp['ase_latents'] = z
        return exp


class _Space:
    def __init__(self, n, low=None, high=None):
        import numpy as np
        self.shape = (n,)
        self.low = np.full(n, -1.0 if low is None else low, dtype=np.float32)
        self.high = np.full(n, 1.0 if high is None else high, dtype=np.float32)


class SyntheticVecEnv:
    """Stepping stand-in for the Isaac Gym vectorised task (ase/env/tasks/humanoid_amp.py behind rl_games' vecenv): the
    interface the agents' rollout loop and the players use - ``reset(env_ids)``, ``step(actions)`` ->
    (obs, rewards, dones, infos{'terminate', 'amp_obs'}), ``task.progress_buf / num_envs / get_task_obs_size()``,
    ``fetch_amp_obs_demo(n)`` - with seeded random observations of the right shapes: the dynamics are not the
    point, the data path is.  ``env`` is the object itself (the reference reaches through ``vec_env.env``).
    task_obs_size > 0 appends task observations (the HRL tasks' goal features) to the character observation.
    demo_source: an ``ase_amd.motion_lib.AmpObsDemoSource`` - the demo observations then come from motion clips through the
    device pipeline (HumanoidAMP.fetch_amp_obs_demo, humanoid_amp.py:63-84) instead of the seeded random stream."""

    def __init__(self, spec, seed=0, device='cpu', task_obs_size=0, demo_source=None):
        self.spec, self.device = spec, torch.device(device)
        self.gen = torch.Generator().manual_seed(seed)
        self.task_obs_size = task_obs_size
        self.obs_fs = FeatureScale(spec.obs_size + task_obs_size, self.gen)
        self.amp_fs = FeatureScale(spec.amp_obs_size, self.gen) if spec.amp_obs_size else None
        n = spec.num_envs
        self.observation_space = _Space(spec.obs_size + task_obs_size, low=-float('inf'), high=float('inf'))
        self.action_space = _Space(spec.act_size)
        self.amp_observation_space = _Space(spec.amp_obs_size) if spec.amp_obs_size else None
        self.env = self
        self.task = self
        self.num_envs = n
        self.viewer = None
        self.progress_buf = torch.zeros(n, dtype=torch.int32, device=self.device)
        self._obs = self.obs_fs.draw(n, self.gen).to(self.device)
        self.last_actions = None
        self.steps = 0
        self.demo_source = demo_source
        if demo_source is not None:
            assert demo_source.get_num_amp_obs() == spec.amp_obs_size, "demo source and env disagree on the AMP observation size"

    def get_task_obs_size(self):
        return self.task_obs_size

    def fetch_amp_obs_demo(self, n):
        if self.demo_source is not None:
            return self.demo_source.fetch_amp_obs_demo(n)
        return self.amp_fs.draw(n, self.gen, mean=0.3).to(self.device)

    def reset(self, env_ids=None):
        if env_ids is None:
            ids = torch.arange(self.num_envs)
        else:
            ids = torch.as_tensor(env_ids, dtype=torch.long).cpu().view(-1)
        if len(ids) > 0:
            self._obs[ids.to(self.device)] = self.obs_fs.draw(len(ids), self.gen).to(self.device)
            self.progress_buf[ids.to(self.device)] = 0
        return self._obs.clone()          # a bare tensor, as RLGPUEnv.reset without global observations (ase/run.py:115-133)

    def step(self, actions):
        n = self.num_envs
        self.last_actions = actions
        self.steps += 1
        self._obs = self.obs_fs.draw(n, self.gen).to(self.device)
        self.progress_buf += 1
        dones = (torch.rand(n, generator=self.gen) < 1.0 / self.spec.episode_length)
        terminate = dones & (torch.rand(n, generator=self.gen) < 0.5)
        infos = {'terminate': terminate.to(self.device)}
        if self.amp_fs is not None:
            infos['amp_obs'] = self.amp_fs.draw(n, self.gen).to(self.device)
        rewards = torch.ones(n, device=self.device)                       # humanoid.py:638-642: task reward == 1
        return self._obs.clone(), rewards, dones.to(torch.uint8).to(self.device), infos
