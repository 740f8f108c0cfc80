"""Players with the reference's plugin interface, running on the HIP inference path.

Mirrors (paths under /root/reference/ase/):
  CommonPlayer         learning/common_player.py:10-187   (rl_games PpoPlayerContinuous / BasePlayer)
  AMPPlayerContinuous  learning/amp_players.py:8-83
  ASEPlayer            learning/ase_players.py:9-152
  HRLPlayer            learning/hrl_players.py:18-260

Same constructor ``Player(config)`` (the yaml ``params.config`` block + ``network`` + ``env_info`` / ``vec_env``; the ``player``
sub-block holds ``games_num / deterministic / print_stats``), same public methods (``restore``, ``run``, ``get_action``,
``env_reset``, ``env_step``, ``_preproc_obs``, ``_eval_disc``, ``_calc_amp_rewards`` ...).  What differs: the network is
never called through autograd modules - observation normalisation, the MLPs, the Normal sample and (ASE) the latent draw
are launches of libase_hip.so (``UpdateEngine.policy_act`` / ``amp_heads``).  Viewer / rendering hooks are out of scope
(Isaac Gym): ``_post_step`` and ``_change_char_color`` are no-ops.
"""
import copy
import os
import time

import numpy as np
import torch

from .agents import rescale_actions


class CommonPlayer:
    kind = 'ppo'

    def __init__(self, config):
        self.config = config
        pc = config.get('player', {}) or {}
        self.env = config.get('vec_env', None)
        self.env_info = config.get('env_info') or self._env_info_from_env()
        self.observation_space = self.env_info['observation_space']
        self.action_space = self.env_info['action_space']
        self.obs_shape = tuple(self.observation_space.shape)
        self.num_agents = self.env_info.get('agents', 1)
        self.value_size = self.env_info.get('value_size', 1)
        self.device = torch.device(config.get('device', config.get('device_name', 'cuda:0')))
        self.games_num = pc.get('games_num', 2000)
        self.is_determenistic = pc.get('determenistic', pc.get('deterministic', True))       # (rl_games' spelling)
        self.n_game_life = pc.get('n_game_life', 1)
        self.print_stats = pc.get('print_stats', True)
        self.render_env = pc.get('render', False)
        self.render_sleep = pc.get('render_sleep', 0.002)
        self.max_steps = pc.get('max_steps', 108000 // 4)
        self.is_tensor_obses = True
        self.states = None
        self.network = config['network']
        self._setup_action_space()
        self.mask = [False]
        self.normalize_input = config['normalize_input']
        self._build_net(self._build_net_config())

    # ------------------------------------------------------------------ construction
    def _env_info_from_env(self):
        e = self.env
        info = {'observation_space': e.observation_space, 'action_space': e.action_space}
        if hasattr(e, 'amp_observation_space'):
            info['amp_observation_space'] = e.amp_observation_space
        return info

    def _setup_action_space(self):
        self.actions_num = self.action_space.shape[0]
        low, high = getattr(self.action_space, 'low', None), getattr(self.action_space, 'high', None)
        n = self.actions_num
        self.actions_low = torch.as_tensor(np.full(n, -1.0) if low is None else np.asarray(low), dtype=torch.float32,
                                           device=self.device)
        self.actions_high = torch.as_tensor(np.full(n, 1.0) if high is None else np.asarray(high), dtype=torch.float32,
                                            device=self.device)

    def _build_net_config(self):
        return {'actions_num': self.actions_num, 'input_shape': self.obs_shape, 'num_seqs': self.num_agents,
                'device': self.device}

    def _engine_cfg(self):
        """Flags of the forward-only engine behind the network (ase_amd.inference.InferenceEngine)."""
        from ..inference import _INFER_CFG
        c = dict(_INFER_CFG)
        c.update(normalize_input=self.normalize_input, normalize_value=self.config.get('normalize_value', False),
                 normalize_amp_input=self.config.get('normalize_amp_input', True), seed=self.config.get('seed', 0))
        return c

    def _build_net(self, config):
        from ..engine import UpdateEngine
        from ..inference import InferenceEngine
        self.model = self.network.build(config)
        self.model.to(self.device)
        self.is_rnn = False
        net = self.model.a2c_network
        backend = self.config.get('backend', None)
        if backend is None:
            from ..backend import HipBackend
            backend = HipBackend(self.device, x3=(self.config.get('precision') == 'bf16x3'))
        self.backend = backend
        # the trainer's resolver: f16gp32 / f16gpx3 checkpoints play in f16, mixed_precision without a precision key = f16
        from ..cfg import resolve_precision
        self.precision, dtype = resolve_precision(self.config)
        self.engine = UpdateEngine(net.kind, net, self._engine_cfg(), backend, minibatch=0, amp_minibatch=0, dtype=dtype)
        net.infer = InferenceEngine(net, self.engine)
        self.action_rng = torch.tensor([int(self.config.get('seed', 0)) ^ 0x91A7E5, 0], dtype=torch.int64, device=self.device)

    # ------------------------------------------------------------------ checkpoints (rl_games BasePlayer.restore)
    @staticmethod
    def _load(fn):
        return fn if isinstance(fn, dict) else torch.load(fn, map_location='cpu', weights_only=False)

    @staticmethod
    def _set_rms(vec, sd):
        D = (vec.numel() - 1) // 2
        vec[:D] = sd['running_mean'].to(vec.device).view(-1)
        vec[D:2 * D] = sd['running_var'].to(vec.device).view(-1)
        vec[2 * D] = sd['count'].to(vec.device)

    def restore(self, fn):
        ck = self._load(fn)
        self.model.load_state_dict(ck['model'])
        if self.normalize_input and 'running_mean_std' in ck:
            self._set_rms(self.engine.obs_state, ck['running_mean_std'])
        if self.config.get('normalize_value', False) and 'reward_mean_std' in ck:
            self._set_rms(self.engine.val_state, ck['reward_mean_std'])
        self.engine.refresh_shadows()
        return ck

    # ------------------------------------------------------------------ acting
    def _preproc_obs(self, obs_batch):
        if obs_batch.dtype == torch.uint8:
            obs_batch = obs_batch.float() / 255.0
        if not self.normalize_input:
            return obs_batch
        e = self.engine
        x = obs_batch.reshape(-1, obs_batch.shape[-1]).contiguous().float()
        out = torch.empty_like(x)
        mean, std = e._eval_stats(e.obs_state, e.obs, 'obs')
        self.backend.rms_normalize(x, e.obs, None, (0, 0), x.shape[0], mean, std, [out])
        return out.view(obs_batch.shape)

    def _act(self, obs, z=None):
        if obs.dim() == len(self.obs_shape):
            obs = obs.unsqueeze(0)
        return self.engine.policy_act(obs.contiguous().float(), z, None, self.action_rng)

    def get_action(self, obs_dict, is_determenistic=False):
        res = self._act(obs_dict['obs'])
        a = res['mus'] if is_determenistic else res['actions']
        return rescale_actions(self.actions_low, self.actions_high, torch.clamp(a, -1.0, 1.0))

    # ------------------------------------------------------------------ environment plumbing
    def obs_to_torch(self, obs):
        if isinstance(obs, dict):
            obs = obs['obs']
        obs = obs if torch.is_tensor(obs) else torch.as_tensor(np.asarray(obs), dtype=torch.float32)
        return {'obs': obs.to(self.device)}

    def env_reset(self, env_ids=None):
        return self.obs_to_torch(self.env.reset(env_ids))

    def env_step(self, env, actions):
        obs, rewards, dones, infos = env.step(actions)
        return self.obs_to_torch(obs), rewards.to(self.device), dones.to(self.device), infos

    def _post_step(self, info):
        pass

    def get_batch_size(self, obs, batch_size):
        return obs.shape[0] if obs.dim() > len(self.obs_shape) else batch_size

    def _play_step(self, obs_dict):
        action = self.get_action(obs_dict, self.is_determenistic)
        return self.env_step(self.env, action)

    def run(self):
        """learning/common_player.py:25-138: play `games_num` episodes, accumulate rewards / steps, print the averages."""
        n_games = self.games_num * self.n_game_life
        sum_rewards = sum_steps = 0.0
        games_played = 0
        for _ in range(n_games):
            if games_played >= n_games:
                break
            obs_dict = self.env_reset()
            batch_size = self.get_batch_size(obs_dict['obs'], 1)
            cr = torch.zeros(batch_size, dtype=torch.float32, device=self.device)
            steps = torch.zeros(batch_size, dtype=torch.float32, device=self.device)
            done_indices = []
            for n in range(self.max_steps):
                obs_dict = self.env_reset(done_indices)
                obs_dict, r, done, info = self._play_step(obs_dict)
                cr += r.view(-1)
                steps += 1
                self._post_step(info)
                if self.render_env:
                    self.env.render(mode='human')
                    time.sleep(self.render_sleep)
                all_done_indices = done.nonzero(as_tuple=False)
                done_indices = all_done_indices[::self.num_agents]
                done_count = len(done_indices)
                games_played += done_count
                if done_count > 0:
                    cur_rewards = cr[done_indices].sum().item()
                    cur_steps = steps[done_indices].sum().item()
                    cr = cr * (1.0 - done.float())
                    steps = steps * (1.0 - done.float())
                    sum_rewards += cur_rewards
                    sum_steps += cur_steps
                    if self.print_stats:
                        print('reward:', cur_rewards / done_count, 'steps:', cur_steps / done_count)
                    if batch_size // self.num_agents == 1 or games_played >= n_games:
                        break
                done_indices = done_indices[:, 0]
        self.games_played, self.sum_rewards, self.sum_steps = games_played, sum_rewards, sum_steps
        if self.print_stats and games_played:
            print('av reward:', sum_rewards / games_played * self.n_game_life, 'av steps:',
                  sum_steps / games_played * self.n_game_life)


class AMPPlayerContinuous(CommonPlayer):
    kind = 'amp'

    def __init__(self, config):
        self._normalize_amp_input = config.get('normalize_amp_input', True)
        self._disc_reward_scale = config['disc_reward_scale']
        super().__init__(config)

    def restore(self, fn):
        if isinstance(fn, str) and fn == 'Base':
            return None
        ck = super().restore(fn)
        if self._normalize_amp_input:
            self._set_rms(self.engine.amp_state, ck['amp_input_mean_std'])
        return ck

    def _build_net_config(self):
        config = super()._build_net_config()
        sp = self.env.amp_observation_space if hasattr(self.env, 'amp_observation_space') else \
            self.env_info['amp_observation_space']
        config['amp_input_shape'] = tuple(getattr(sp, 'shape', sp))
        return config

    def _preproc_amp_obs(self, amp_obs):
        if not self._normalize_amp_input:
            return amp_obs
        e = self.engine
        x = amp_obs.reshape(-1, amp_obs.shape[-1]).contiguous().float()
        out = torch.empty_like(x)
        mean, std = e._eval_stats(e.amp_state, e.amp, 'amp')
        self.backend.rms_normalize(x, e.amp, None, (0, 0), x.shape[0], mean, std, [out])
        return out.view(amp_obs.shape)

    def _eval_disc(self, amp_obs):
        HD, _ = self.engine.amp_heads(amp_obs.reshape(-1, amp_obs.shape[-1]).contiguous().float())
        return HD[:, 0:1].clone()

    def _calc_disc_rewards(self, amp_obs):
        HD, _ = self.engine.amp_heads(amp_obs.reshape(-1, amp_obs.shape[-1]).contiguous().float())
        r = torch.empty(HD.shape[0], 1, dtype=torch.float32, device=self.device)
        self.backend.disc_reward(HD, r, HD.shape[0], self._disc_reward_scale)
        return r.view(*amp_obs.shape[:-1], 1)

    def _calc_amp_rewards(self, amp_obs):
        return {'disc_rewards': self._calc_disc_rewards(amp_obs)}


class ASEPlayer(AMPPlayerContinuous):
    kind = 'ase'

    def __init__(self, config):
        self._latent_dim = config['latent_dim']
        self._latent_steps_min = config.get('latent_steps_min', np.inf)
        self._latent_steps_max = config.get('latent_steps_max', np.inf)
        self._enc_reward_scale = config['enc_reward_scale']
        super().__init__(config)
        if self.env is not None and hasattr(self.env, 'task'):
            batch_size = self.env.task.num_envs
        else:
            batch_size = self.env_info['num_envs']
        self._ase_latents = torch.zeros(batch_size, self._latent_dim, dtype=torch.float32, device=self.device)
        self._np_rng = np.random.RandomState(int(config.get('seed', 0)))
        self._latent_step_count = 0

    def _build_net_config(self):
        config = super()._build_net_config()
        config['ase_latent_shape'] = (self._latent_dim,)
        return config

    def run(self):
        self._reset_latent_step_count()
        super().run()

    def get_action(self, obs_dict, is_determenistic=False):
        self._update_latents()
        res = self._act(obs_dict['obs'], self._ase_latents)
        a = res['mus'] if is_determenistic else res['actions']
        return rescale_actions(self.actions_low, self.actions_high, torch.clamp(a, -1.0, 1.0))

    def env_reset(self, env_ids=None):
        obs = super().env_reset(env_ids)
        self._reset_latents(env_ids)
        return obs

    def _reset_latents(self, done_env_ids=None):
        if done_env_ids is None:
            done_env_ids = torch.arange(self._ase_latents.shape[0], dtype=torch.long, device=self.device)
        done_env_ids = torch.as_tensor(done_env_ids, dtype=torch.long, device=self.device)
        if len(done_env_ids) == 0:
            return
        self._ase_latents[done_env_ids] = self.model.a2c_network.sample_latents(len(done_env_ids))
        self._change_char_color(done_env_ids)

    def _update_latents(self):
        if self._latent_step_count <= 0:
            self._reset_latents()
            self._reset_latent_step_count()
        else:
            self._latent_step_count -= 1

    def _reset_latent_step_count(self):
        self._latent_step_count = self._np_rng.randint(self._latent_steps_min, self._latent_steps_max)

    def _eval_enc(self, amp_obs):
        _, e = self.engine.amp_heads(amp_obs.reshape(-1, amp_obs.shape[-1]).contiguous().float())
        e = e[:, :self._latent_dim]
        return e / e.norm(dim=-1, keepdim=True).clamp_min(1e-12)

    def _calc_enc_rewards(self, amp_obs, ase_latents):
        _, enc = self.engine.amp_heads(amp_obs.reshape(-1, amp_obs.shape[-1]).contiguous().float())
        n = enc.shape[0]
        r = torch.empty(n, 1, dtype=torch.float32, device=self.device)
        self.backend.enc_reward(enc, ase_latents.reshape(n, -1).contiguous(), r, n, self._latent_dim, self._enc_reward_scale)
        return r.view(*amp_obs.shape[:-1], 1)

    def _calc_amp_rewards(self, amp_obs, ase_latents):
        return {'disc_rewards': self._calc_disc_rewards(amp_obs), 'enc_rewards': self._calc_enc_rewards(amp_obs, ase_latents)}

    def _change_char_color(self, env_ids):
        pass                       # viewer only (learning/ase_players.py:141-152)


class HRLPlayer(CommonPlayer):
    kind = 'ppo'

    def __init__(self, config):
        llc = config['llc_config']
        if not isinstance(llc, dict):
            import yaml
            with open(os.path.join(os.getcwd(), llc), 'r') as f:
                llc = yaml.load(f, Loader=yaml.SafeLoader)
        self._llc_params = llc['params']
        self._latent_dim = self._llc_params['config']['latent_dim']
        super().__init__(config)
        self._task_size = self.env.task.get_task_obs_size()
        self._llc_steps = config['llc_steps']
        llc_checkpoint = config['llc_checkpoint']
        assert llc_checkpoint != ""                                          # learning/hrl_players.py:30
        self._build_llc(self._llc_params, llc_checkpoint)

    def _setup_action_space(self):
        super()._setup_action_space()
        self.actions_num = self._latent_dim

    def get_action(self, obs_dict, is_determenistic=False):
        res = self._act(obs_dict['obs'])
        a = res['mus'] if is_determenistic else res['actions']
        return torch.clamp(a, -1.0, 1.0)                                     # learning/hrl_players.py:58-60

    def _play_step(self, obs_dict):
        action = self.get_action(obs_dict, self.is_determenistic)
        return self.env_step(self.env, obs_dict, action)

    def env_step(self, env, obs_dict, action):
        obs = obs_dict['obs']
        rewards = done_count = disc_rewards = 0.0
        for t in range(self._llc_steps):
            llc_actions = self._compute_llc_action(obs, action)
            obs, curr_rewards, curr_dones, infos = env.step(llc_actions)
            obs = (obs['obs'] if isinstance(obs, dict) else obs).to(self.device)
            rewards = rewards + curr_rewards.to(self.device)
            done_count = done_count + curr_dones.to(self.device).float()
            disc_rewards = disc_rewards + self._calc_disc_reward(infos['amp_obs'].to(self.device))
        rewards = rewards / self._llc_steps
        dones = (done_count > 0).to(done_count.dtype)
        infos['disc_rewards'] = disc_rewards / self._llc_steps
        return self.obs_to_torch(obs), rewards, dones, infos

    def _build_llc(self, config_params, checkpoint):
        from . import models
        from .network_builder import ASEBuilder
        import types
        builder = ASEBuilder()
        builder.load(config_params['network'])
        env_info = copy.copy(self.env_info)
        obs_size = self.obs_shape[0] - self._task_size
        env_info['observation_space'] = types.SimpleNamespace(shape=(obs_size,))
        sp = self.env.amp_observation_space
        env_info['amp_observation_space'] = types.SimpleNamespace(shape=tuple(getattr(sp, 'shape', sp)))
        env_info['num_envs'] = self.env.task.num_envs
        cfg = copy.copy(config_params['config'])
        cfg.update(network=models.ModelASEContinuous(builder), env_info=env_info, device=self.device,
                   precision=self.config.get('precision', 'bf16'), seed=self.config.get('seed', 0), vec_env=None)
        if self.config.get('backend') is not None:
            cfg['backend'] = self.config['backend']
        self._llc_agent = ASEPlayer(cfg)
        self._llc_agent.restore(checkpoint)

    def _extract_llc_obs(self, obs):
        return obs[..., :obs.shape[-1] - self._task_size]

    def _compute_llc_action(self, obs, actions):
        llc = self._llc_agent
        e = llc.engine
        z = e._scr('hrl_z', actions.shape[0], self._latent_dim, torch.float32)
        self.backend.normalize_rows(actions.contiguous(), z, actions.shape[0], self._latent_dim)
        mu = e.policy_forward(self._extract_llc_obs(obs).contiguous(), z, want=('mu',))['mu']
        return rescale_actions(self.actions_low, self.actions_high, torch.clamp(mu, -1.0, 1.0))

    def _calc_disc_reward(self, amp_obs):
        return self._llc_agent._calc_disc_rewards(amp_obs)
