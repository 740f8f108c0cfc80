"""Agents with the reference's plugin interface for the update path.

Mirrors (paths under /root/reference/ase/):
  CommonAgent  learning/common_agent.py:25-564   plain PPO (also the HRL high-level update)
  AMPAgent     learning/amp_agent.py:21-628      + discriminator, demo / replay buffers
  ASEAgent     learning/ase_agent.py:12-538      + latents, encoder, diversity

Same constructor signature ``Agent(base_name, config)`` where ``config`` is the yaml ``params.config``
block plus the keys rl_games' Runner injects (``network``, ``env_info``, ``num_actors``; see
learning/hrl_agent.py:223-229), same public methods (``train_epoch``, ``play_steps``,
``prepare_dataset``, ``calc_gradients``, ``train_actor_critic``, ``discount_values``,
``set_eval / set_train``, ``get/set_stats_weights``, ``get/set_full_state_weights``, ``save / restore``),
same ``train_result`` / ``train_info`` keys and the same checkpoint dictionary.

What differs is where the work happens: everything between the filled experience buffer and the end
of ``train_epoch`` runs in libase_hip.so through ``UpdateEngine`` (optionally replayed from a hipGraph),
minibatches are addressed by index instead of being gathered into copies, and per-step scalars stay
on the device until the epoch ends (the reference's ``kl.item()`` per step, amp_agent.py:227, is a no-op
for its IdentityScheduler and is dropped).

Isaac Gym is out of scope (BASELINE.json): ``config['vec_env']`` is any object with
``experience(policy) -> dict`` and ``fetch_amp_obs_demo(n)`` (ase_amd.synthetic.SyntheticSource).
"""
import os
import time

import numpy as np
import torch

from .. import lib as L
from ..engine import UpdateEngine
from ..inference import InferenceEngine
from .replay_buffer import ReplayBuffer


class AverageMeter:
    """rl_games torch_ext.AverageMeter: mean of the last `max_size` values pushed (episode rewards / lengths)."""

    def __init__(self, in_shape, max_size, device='cpu'):
        self.max_size = max_size
        self.current_size = 0
        self.mean = torch.zeros(in_shape, dtype=torch.float32, device=device)

    def update(self, values):
        size = values.size()[0]
        if size == 0:
            return
        new_mean = torch.mean(values.float(), dim=0)
        size = int(np.clip(size, 0, self.max_size))
        old_size = min(self.max_size - size, self.current_size)
        size_sum = old_size + size
        self.current_size = size_sum
        self.mean = (self.mean * old_size + new_mean.to(self.mean.device) * size) / size_sum

    def clear(self):
        self.current_size = 0
        self.mean.fill_(0)

    def get_mean(self):
        return self.mean.squeeze(0).cpu().numpy()


class _NullObserver:
    """rl_games AlgoObserver interface (config['features']['observer']); the default does nothing."""

    def after_init(self, algo):
        pass

    def process_infos(self, infos, done_indices):
        pass

    def after_steps(self):
        pass

    def after_print_stats(self, frame, epoch_num, total_time):
        pass


class ScalarLog:
    """Stand-in for tensorboardX.SummaryWriter (logging back ends are out of scope): keeps the last value per tag."""

    def __init__(self):
        self.scalars = {}

    def add_scalar(self, tag, value, step=None):
        self.scalars[tag] = (float(value), step)


def rescale_actions(low, high, action):
    """rl_games players.rescale_actions"""
    d = (high - low) / 2.0
    m = (high + low) / 2.0
    return action * d + m


def _mean_list(vals):
    return torch.stack([torch.as_tensor(v, dtype=torch.float32).reshape(-1).mean() for v in vals]).mean()


class CommonAgent:
    kind = 'ppo'

    def __init__(self, base_name, config):
        self.base_name = base_name
        self.config = config
        self.env_info = config['env_info']
        self.ppo_device = torch.device(config.get('device', 'cuda:0'))
        self.num_actors = config['num_actors']
        self.num_agents = self.env_info.get('agents', 1)
        self.horizon_length = config['horizon_length']
        self.minibatch_size = config['minibatch_size']
        self.mini_epochs_num = config['mini_epochs']
        self.batch_size = self.horizon_length * self.num_actors * self.num_agents
        assert self.batch_size % self.minibatch_size == 0            # rl_games A2CBase
        self.num_minibatches = self.batch_size // self.minibatch_size
        self.normalize_input = config['normalize_input']
        self.normalize_value = config.get('normalize_value', False)
        self.normalize_advantage = config['normalize_advantage']
        self.e_clip, self.clip_value = config['e_clip'], config['clip_value']
        self.critic_coef, self.entropy_coef = config['critic_coef'], config['entropy_coef']
        self.gamma, self.tau = config['gamma'], config['tau']
        self.bounds_loss_coef = config.get('bounds_loss_coef', None)
        self.truncate_grads = config.get('truncate_grads', False)       # global-norm clip (UpdateEngine.truncate)
        self.grad_norm = config.get('grad_norm', 1.0)
        # mixed_precision (rl_games: torch.cuda.amp autocast = half arithmetic + GradScaler, learning/ase_agent.py:216,271-288)
        # selects the half-storage mode 'f16' below WITH the GradScaler's behaviour (config loss_scale: 'dynamic' - overflow
        # detection, skipped steps, backoff / growth of the scale: UpdateEngine.scaler_update); an explicitly named precision mode
        # ('precision': 'f16' / 'f16gp32' / 'f16gpx3') keeps the static power-of-two scale unless loss_scale says otherwise
        # lr_schedule: constant | adaptive (rl_games AdaptiveScheduler on every step's kl, schedule_type 'legacy' - the default;
        # learning/common_agent.py:204-208).  The per-mini-epoch / per-epoch variants ('standard', 'standard_epoch') are not built.
        assert config.get('lr_schedule', 'constant') in ('constant', 'adaptive', None)
        assert config.get('lr_schedule', 'constant') != 'adaptive' or config.get('schedule_type', 'legacy') == 'legacy'
        self.multi_gpu = config.get('multi_gpu', False)
        self.world_size, self.rank = config.get('world_size', 1), config.get('rank', 0)
        # 'shard' (strong scaling: the R-rank update equals the 1-rank update) | 'horovod' (the reference's semantics:
        # every rank owns its environments, gradients averaged, statistics averaged per epoch)
        self.dp_mode = config.get('dp_mode', 'shard')
        self.multi_gpu = self.multi_gpu or self.world_size > 1
        self.seed = int(config.get('seed', 0) or 0)
        self.last_lr = float(config['learning_rate'])
        self.epoch_num = 0
        self.frame = 0
        self.last_mean_rewards = -100500
        self.is_rnn = False
        self.obs_shape = self.env_info['observation_space'].shape
        self.actions_num = self.env_info['action_space'].shape[0]
        self.value_size = self.env_info.get('value_size', 1)
        self.vec_env = config.get('vec_env', None)
        self.name = config.get('name', base_name)
        self.max_epochs = config.get('max_epochs', 1e6)
        self.save_freq = config.get('save_frequency', 0)
        self.print_stats = config.get('print_stats', True) and self.rank == 0
        self.games_to_track = config.get('games_to_track', 100)
        self.clip_actions = config.get('clip_actions', True)
        self._save_intermediate = config.get('save_intermediate', False)
        self.nn_dir = config.get('train_dir', os.path.join('runs', self.name, 'nn'))
        rs = config.get('reward_shaper', {}) or {}
        self._reward_scale, self._reward_shift = float(rs.get('scale_value', 1.0)), float(rs.get('shift_value', 0.0))
        self.algo_observer = (config.get('features', {}) or {}).get('observer', None) or _NullObserver()
        self.writer = config.get('writer', None) or ScalarLog()
        self.is_tensor_obses = True
        self._load_config_params(config)
        self._setup_action_space()

        self.network = config['network']
        self.model = self.network.build(self._build_net_config())
        self.model.to(self.ppo_device)
        # precision: 'bf16' (bf16 storage + MFMA, throughput mode) | 'f32' (exact f32 MFMA) |
        #            'bf16x3' (f32 storage, every product as three bf16 MFMAs on a hi/lo split: f32-grade results)
        #            'f16' (IEEE half storage + MFMA, f32 accumulate, static gradient scale: same rate as bf16, 3 more
        #                   mantissa bits - the reference's mixed_precision=True arithmetic)
        #            'f16gp32' f16 with the gradient penalty's VALUE path (demo-row forward for exact masks, chain) in exact f32:
        #                   the penalty is a cancelling sum in the discriminator's weights and the one loss scalar half storage
        #                   does not hold to 1e-4 at every training state; its backward stays in half
        from ..cfg import resolve_precision
        precision, dtype = resolve_precision(config)
        if precision in ('f16gp32', 'f16gpx3'):      # (f16gpx3: the same path with three f16 MFMAs per product on hi / lo half splits)
            config = self.config = dict(config, gp_f32=True if precision == 'f16gp32' else 'x3')
        self.precision = precision
        backend = config.get('backend', None)
        if backend is None:
            from ..backend import HipBackend
            backend = HipBackend(self.ppo_device, x3=(precision == 'bf16x3'))
        self.backend = backend
        self.engine = UpdateEngine(self.kind, self.model.a2c_network, config, backend, minibatch=self.minibatch_size,
                                   amp_minibatch=getattr(self, '_amp_minibatch_size', 0), dtype=dtype,
                                   world_size=self.world_size, rank=self.rank, dp_mode=self.dp_mode,
                                   grad_scale=config.get('grad_scale', None))
        self.model.a2c_network.infer = InferenceEngine(self.model.a2c_network, self.engine)
        self.use_graph = bool(config.get('graph_capture', False))
        self._snapshot_aside = bool(config.get('snapshot_aside', True))     # per-step result snapshots on a side stream
        self._snapshot_stream = None
        # Result rings (default): every optimisation step of an update writes its train_result scalars and discriminator logits
        # into its own slot of a device ring, read ONCE at the end of the update - no per-step snapshot copies between two steps.
        # (A captured hipGraph binds one slot per minibatch position: that mode keeps the per-step snapshots.)
        self._use_rings = bool(config.get('result_rings', True)) and config.get('graph_capture') != 'hipgraph'
        # the update's own stream (see update()); a captured hipGraph and the CPU emulator keep the caller's
        prio = int(config.get('main_stream_priority', -1))
        self._main_stream = torch.cuda.Stream(device=self.ppo_device, priority=prio) \
            if (prio != 0 and getattr(backend, 'name', '') == 'hip' and config.get('graph_capture') != 'hipgraph') else None
        if self._use_rings:
            self.engine.set_result_slots(self.mini_epochs_num * self.num_minibatches)
        self._graphs = {}
        self._train_mode = True
        self.train_result = None
        # Every random draw of the update (dataset permutations, ring sample permutations, replay keep masks) comes from
        # ONE generator.  Sharded data parallel: all ranks seed it identically and make the same draws in the same order,
        # so minibatch r of rank k is the k-th shard of the same global minibatch; Horovod mode: rank-distinct streams.
        self._gen = torch.Generator(device=self.ppo_device)
        self._gen.manual_seed(self.seed * 1000003 + 12345 + (0 if self.dp_mode == 'shard' else 7919 * self.rank))
        # draws whose RESULT the host needs (the replay ring's keep mask decides how many rows are stored) are made on the host
        # and uploaded asynchronously: no update ends in a device -> host read-back (_store_replay_amp_obs)
        self._host_gen = torch.Generator()
        self._host_gen.manual_seed(self.seed * 1000003 + 54321 + (0 if self.dp_mode == 'shard' else 7919 * self.rank))
        self._upload_ring, self._upload_pos = [], 0
        self.dataset_perm = self._randperm(self.batch_size)
        self.action_rng = torch.tensor([(self.seed ^ 0xAC7105) + (0 if self.dp_mode == 'shard' else self.rank), 0],
                                       dtype=torch.int64, device=self.ppo_device)       # Philox stream of the rollout's actions
        self.game_rewards = AverageMeter(self.value_size, self.games_to_track)
        self.game_lengths = AverageMeter(1, self.games_to_track)
        self.obs = None
        self.init_tensors()
        self.algo_observer.after_init(self)
        if self.world_size > 1:
            self._sync_initial_state()

    def _randperm(self, n):
        return torch.randperm(n, device=self.ppo_device, generator=self._gen).to(torch.int32)

    def _upload_i32(self, t_cpu, cap):
        """Host int32 vector -> device without blocking the host: pinned staging + device buffers in a ring of 8 (the host may
        run several updates ahead of the GPU; a slot is reused only after the copy that last used it has completed, and its
        consumer - the ring store of that update - was enqueued on the same stream right behind that copy)."""
        n = int(t_cpu.numel())
        if torch.device(self.ppo_device).type != 'cuda':
            return t_cpu.to(self.ppo_device)
        if not self._upload_ring:
            for _ in range(8):
                self._upload_ring.append((torch.empty(cap, dtype=torch.int32).pin_memory(),
                                          torch.empty(cap, dtype=torch.int32, device=self.ppo_device), torch.cuda.Event()))
        host, dev, ev = self._upload_ring[self._upload_pos % len(self._upload_ring)]
        self._upload_pos += 1
        ev.synchronize()
        host[:n].copy_(t_cpu)
        dev[:n].copy_(host[:n], non_blocking=True)
        ev.record()
        return dev[:n]

    def _sync_initial_state(self):
        """rl_games HorovodWrapper.setup_algo: rank 0's parameters / optimizer state / statistics everywhere."""
        self.engine.sync_from_rank0()

    def sync_stats(self):
        """rl_games HorovodWrapper.sync_stats (learning/common_agent.py:106-107): average every running-statistics buffer
        over the ranks, sum the frame counters.  Sharded mode keeps the statistics identical by construction."""
        if self.world_size <= 1 or self.dp_mode == 'shard':
            return
        e = self.engine
        bufs = [e.obs_state, e.val_state] + ([e.amp_state] if e.has_disc else [])
        for t in bufs:
            e._ar(t)
            t.mul_(1.0 / self.world_size)
        cf = torch.tensor([float(self.curr_frames)], dtype=torch.float64, device=self.ppo_device)
        e._ar(cf)
        self.curr_frames = int(cf.item())

    def _setup_action_space(self):
        sp = self.env_info['action_space']
        low, high = getattr(sp, 'low', None), getattr(sp, 'high', None)
        n = sp.shape[0]
        self.actions_low = torch.as_tensor(np.full(n, -1.0) if low is None else np.asarray(low), dtype=torch.float32,
                                           device=self.ppo_device)
        self.actions_high = torch.as_tensor(np.full(n, 1.0) if high is None else np.asarray(high), dtype=torch.float32,
                                            device=self.ppo_device)

    # ------------------------------------------------------------------ config
    def _load_config_params(self, config):
        pass

    def _build_net_config(self):
        return {'actions_num': self.actions_num, 'input_shape': self.obs_shape,
                'num_seqs': self.num_actors * self.num_agents, 'value_size': self.value_size,
                'device': self.ppo_device}

    # ------------------------------------------------------------------ buffers
    def _drop_graphs(self):
        """Recorded launch programs / captured graphs have the addresses of the experience and dataset buffers baked in:
        whenever those are re-allocated the recordings go (and the library-side programs are freed)."""
        for g in getattr(self, '_graphs', {}).values():
            if not g['hipgraph']:
                for prog in g['graphs']:
                    self.backend.prog_destroy(prog)
        self._graphs = {}

    def init_tensors(self):
        H, N, dev = self.horizon_length, self.num_actors * self.num_agents, self.ppo_device
        f32 = dict(dtype=torch.float32, device=dev)
        A = self.actions_num
        self._drop_graphs()
        self.experience = {            # rl_games ExperienceBuffer.tensor_dict layout (time-major)
            'obses': torch.zeros(H, N, *self.obs_shape, **f32), 'rewards': torch.zeros(H, N, 1, **f32),
            'values': torch.zeros(H, N, 1, **f32), 'neglogpacs': torch.zeros(H, N, **f32),
            'dones': torch.zeros(H, N, dtype=torch.uint8, device=dev), 'actions': torch.zeros(H, N, A, **f32),
            'mus': torch.zeros(H, N, A, **f32), 'sigmas': torch.zeros(H, N, A, **f32),
            'next_obses': torch.zeros(H, N, *self.obs_shape, **f32), 'next_values': torch.zeros(H, N, 1, **f32)}
        self.tensor_list = ['actions', 'neglogpacs', 'values', 'mus', 'sigmas', 'obses', 'states', 'dones', 'next_obses']
        self.current_rewards = torch.zeros(N, self.value_size, dtype=torch.float32, device=dev)   # rl_games A2CBase.init_tensors
        self.current_lengths = torch.zeros(N, dtype=torch.float32, device=dev)
        self.dones = torch.ones(N, dtype=torch.uint8, device=dev)

    # ------------------------------------------------------------------ mode switches / stats
    def set_eval(self):
        self._train_mode = False

    def set_train(self):
        self._train_mode = True

    @staticmethod
    def _rms_to_state_dict(vec):
        D = (vec.numel() - 1) // 2
        return {'running_mean': vec[:D].clone(), 'running_var': vec[D:2 * D].clone(), 'count': vec[2 * D].clone()}

    @staticmethod
    def _rms_from_state_dict(vec, sd):
        D = (vec.numel() - 1) // 2
        vec[:D] = sd['running_mean'].to(vec.device).view(-1)
        vec[D:2 * D] = sd['running_var'].to(vec.device).view(-1)
        vec[2 * D] = sd['count'].to(vec.device)

    def get_stats_weights(self):
        state = {}
        if self.normalize_input:
            state['running_mean_std'] = self._rms_to_state_dict(self.engine.obs_state)
        if self.normalize_value:
            state['reward_mean_std'] = self._rms_to_state_dict(self.engine.val_state)
        return state

    def set_stats_weights(self, weights):
        if self.normalize_input:
            self._rms_from_state_dict(self.engine.obs_state, weights['running_mean_std'])
        if self.normalize_value:
            self._rms_from_state_dict(self.engine.val_state, weights['reward_mean_std'])

    def get_weights(self):
        state = self.get_stats_weights()
        state['model'] = self.model.state_dict()
        return state

    def set_weights(self, weights):
        self.model.load_state_dict(weights['model'])
        self.set_stats_weights(weights)
        self.engine.refresh_shadows()

    def _optimizer_state_dict(self):
        """torch.optim.Adam.state_dict() layout over model.parameters() order (what the reference saves)."""
        e, net = self.engine, self.model.a2c_network
        state, ids = {}, []
        step = torch.tensor(float(e.opt_state[0].item()))
        for i, (k, p) in enumerate(self.model.named_parameters()):
            ids.append(i)
            name = k.replace('a2c_network.', '', 1)
            o, shp = net.param_slices[name]
            n = int(np.prod(shp))
            if p.requires_grad and step > 0:
                state[i] = {'step': step.clone(), 'exp_avg': e.adam_m[o:o + n].view(shp).clone(),
                            'exp_avg_sq': e.adam_v[o:o + n].view(shp).clone()}
        group = {'lr': self.last_lr, 'betas': (0.9, 0.999), 'eps': 1e-08, 'weight_decay': 0, 'amsgrad': False,
                 'params': ids}
        return {'state': state, 'param_groups': [group]}

    def _load_optimizer_state_dict(self, sd):
        e, net = self.engine, self.model.a2c_network
        step = 0.0
        for i, (k, p) in enumerate(self.model.named_parameters()):
            st = sd['state'].get(i)
            if st is None:
                continue
            o, shp = net.param_slices[k.replace('a2c_network.', '', 1)]
            n = int(np.prod(shp))
            e.adam_m[o:o + n] = st['exp_avg'].to(e.dev).reshape(-1)
            e.adam_v[o:o + n] = st['exp_avg_sq'].to(e.dev).reshape(-1)
            step = float(st['step'])
        e.opt_state[0] = step
        e.opt_state[1] = float(sd['param_groups'][0]['lr'])
        self.last_lr = float(sd['param_groups'][0]['lr'])          # host mirror (constant AND adaptive schedule)

    def get_full_state_weights(self):
        state = self.get_weights()
        state['epoch'] = self.epoch_num
        state['optimizer'] = self._optimizer_state_dict()
        state['frame'] = self.frame
        state['last_mean_rewards'] = self.last_mean_rewards
        state['env_state'] = None
        # extra key (ignored by the reference's loader): position of the device-side latent / action streams
        state['hip_rng_state'] = {'latents': self.engine.rng_state.cpu().clone() if self.engine.style else None,
                                  'diversity': self.engine.div_rng.cpu().clone() if self.engine.style else None,
                                  'actions': self.action_rng.cpu().clone()}
        return state

    def set_full_state_weights(self, weights):
        self.set_weights(weights)
        self.epoch_num = weights['epoch']
        self._load_optimizer_state_dict(weights['optimizer'])
        self.frame = weights.get('frame', 0)
        self.last_mean_rewards = weights.get('last_mean_rewards', -100500)
        rs = weights.get('hip_rng_state')
        if rs is not None:
            if rs.get('latents') is not None and self.engine.style:
                self.engine.rng_state.copy_(rs['latents'])
                if rs.get('diversity') is not None:
                    self.engine.div_rng.copy_(rs['diversity'])
            self.action_rng.copy_(rs['actions'])
        if self.world_size > 1:
            self._sync_initial_state()

    def save(self, fn):
        torch.save(self.get_full_state_weights(), fn + '.pth')       # rl_games torch_ext.save_checkpoint

    def restore(self, fn):
        self.set_full_state_weights(torch.load(fn, map_location=self.ppo_device, weights_only=False))

    def update_epoch(self):
        self.epoch_num += 1
        return self.epoch_num

    def update_lr(self, lr):
        self.last_lr = lr
        self.engine.opt_state[1] = lr

    # ------------------------------------------------------------------ rollout side (inference path)
    def _policy(self, obs, z=None):
        out = self.engine.policy_forward(obs, z)
        mu = out['mu']
        sigma = torch.exp(mu * 0.0 + self.engine.logstd)
        return mu, sigma, out['value']

    def _preproc_obs(self, obs_batch):
        """rl_games A2CBase._preproc_obs: uint8 -> /255, then the eval-mode observation normaliser.  The agents' own
        rollout path fuses this into the first layer's input buffers (UpdateEngine._policy_nets); this stand-alone form
        serves callers that feed the network-level API (model.a2c_network.eval_actor, learning/hrl_agent.py:231-236)."""
        if obs_batch.dtype == torch.uint8:
            obs_batch = obs_batch.float() / 255.0
        if not self.normalize_input:
            return obs_batch
        e = self.engine
        x = obs_batch.reshape(-1, obs_batch.shape[-1]).contiguous()
        out = torch.empty_like(x, dtype=torch.float32)
        mean, std = e._eval_stats(e.obs_state, e.obs, 'obs')
        self.backend.rms_normalize(x, e.obs, None, (0, 0), x.shape[0], mean, std, [out])
        return out.view(obs_batch.shape)

    def get_action_values(self, obs_dict, *extra):
        """rl_games A2CBase.get_action_values(obs) / AMPAgent(obs, rand_action_probs) / ASEAgent(obs, ase_latents,
        rand_action_probs) (learning/amp_agent.py:139-169, learning/ase_agent.py:117-148): eval-mode model forward, sampled
        action, its neglogp, un-normalised value and - AMP / ASE - the eps-greedy substitution of mu for the rows whose
        Bernoulli(rand_action_probs) draw is 0.  All of it on the device (UpdateEngine.policy_act)."""
        z = probs = None
        if self.kind == 'ase':
            z = extra[0] if len(extra) > 0 else None
            probs = extra[1] if len(extra) > 1 else None
        elif self.kind == 'amp':
            probs = extra[0] if len(extra) > 0 else None
        return self.engine.policy_act(obs_dict['obs'], z, probs, self.action_rng)

    def _eval_critic(self, obs_dict, *extra):
        return self.engine.policy_forward(obs_dict['obs'], *extra[:1], want=('value',))['value']

    # ---- environment plumbing (rl_games A2CBase / ContinuousA2CBase)
    def obs_to_tensors(self, obs):
        if isinstance(obs, dict):
            return {k: (v.to(self.ppo_device) if torch.is_tensor(v) else torch.as_tensor(v, device=self.ppo_device))
                    for k, v in obs.items()}
        return {'obs': obs.to(self.ppo_device) if torch.is_tensor(obs) else torch.as_tensor(obs, device=self.ppo_device)}

    def env_reset(self, env_ids=None):
        return self.obs_to_tensors(self.vec_env.reset(env_ids))

    def preprocess_actions(self, actions):
        if self.clip_actions:
            return rescale_actions(self.actions_low, self.actions_high, torch.clamp(actions, -1.0, 1.0))
        return actions

    def env_step(self, actions):
        obs, rewards, dones, infos = self.vec_env.step(self.preprocess_actions(actions))
        if self.value_size == 1:
            rewards = rewards.unsqueeze(1)
        return self.obs_to_tensors(obs), rewards.to(self.ppo_device), dones.to(self.ppo_device), infos

    def rewards_shaper(self, rewards):
        return (rewards + self._reward_shift) * self._reward_scale

    def _rollout_extras(self, n, res_dict, infos):
        """Per-step experience fields beyond CommonAgent's (AMP: amp_obs, rand_action_mask; ASE: + ase_latents)."""

    def _act(self):
        return self.get_action_values(self.obs)

    def _next_values(self):
        return self._eval_critic(self.obs)

    def _before_act(self):
        pass

    def play_steps(self):
        """learning/common_agent.py:241-307 (amp_agent.py:61-137, ase_agent.py:36-115, hrl_agent.py:95-163): roll the
        vectorised environment for `horizon_length` steps with the HIP inference path, then the tail (rewards, GAE,
        dataset) in UpdateEngine.prepare_epoch.  A source that hands over a whole rollout at once
        (SyntheticSource.experience, the benchmark's generator) is copied in directly."""
        self.set_eval()
        if not hasattr(self.vec_env, 'step'):
            exp = self.vec_env.experience(self._cpu_policy(), **self._experience_kwargs())
            for k, v in exp.items():
                if k in self.experience:
                    self.experience[k].copy_(v.to(self.ppo_device))
            return self._play_steps_tail()
        E = self.experience
        done_indices = []
        if self.obs is None:
            self.obs = self.env_reset()
        for n in range(self.horizon_length):
            self.obs = self.env_reset(done_indices)
            E['obses'][n].copy_(self.obs['obs'])
            self._before_act()
            res = self._act()
            for k in ('actions', 'neglogpacs', 'values', 'mus', 'sigmas'):
                E[k][n].copy_(res[k].view(E[k][n].shape))
            self.obs, rewards, self.dones, infos = self.env_step(res['actions'])
            E['rewards'][n].copy_(self.rewards_shaper(rewards))
            E['next_obses'][n].copy_(self.obs['obs'])
            E['dones'][n].copy_(self.dones)
            self._rollout_extras(n, res, infos)
            terminated = infos['terminate'].float().unsqueeze(-1).to(self.ppo_device)
            E['next_values'][n].copy_(self._next_values() * (1.0 - terminated))
            self.current_rewards += rewards
            self.current_lengths += 1
            all_done_indices = self.dones.nonzero(as_tuple=False)
            done_indices = all_done_indices[::self.num_agents]
            self.game_rewards.update(self.current_rewards[done_indices])
            self.game_lengths.update(self.current_lengths[done_indices])
            self.algo_observer.process_infos(infos, done_indices)
            not_dones = 1.0 - self.dones.float()
            self.current_rewards = self.current_rewards * not_dones.unsqueeze(1)
            self.current_lengths = self.current_lengths * not_dones
            done_indices = done_indices[:, 0]
        return self._play_steps_tail()

    def _experience_kwargs(self):
        return {'with_amp': False, 'with_latents': False}

    def _cpu_policy(self):
        dev = self.ppo_device

        def policy(obs, z):
            mu, sigma, value = self._policy(obs.to(dev), None if z is None else z.to(dev))
            return mu.cpu(), sigma.cpu(), value.cpu()
        return policy

    def _play_steps_tail(self):
        self._ds, self._tail_info, self._remap = self.engine.prepare_epoch(self.experience)
        batch_dict = dict(self._ds)
        batch_dict.update(self._tail_info)
        batch_dict['played_frames'] = self.batch_size
        return batch_dict

    def discount_values(self, mb_fdones, mb_values, mb_rewards, mb_next_values):
        """GAE with the reference's signature (learning/common_agent.py:437-449), on the HIP kernel."""
        H, N = mb_rewards.shape[0], mb_rewards.shape[1]
        advs, rets = torch.empty_like(mb_rewards), torch.empty_like(mb_rewards)
        self.backend.gae(mb_fdones.to(torch.uint8).contiguous(), mb_values.contiguous(), mb_next_values.contiguous(),
                         mb_rewards.contiguous(), None, None, 1.0, 0.0, 0.0, self.gamma, self.tau, advs, rets, H, N)
        return advs

    # ------------------------------------------------------------------ update side
    def prepare_dataset(self, batch_dict):
        """Kept for interface parity; the dataset (advantages, normalised values / returns) is produced by
        UpdateEngine.prepare_epoch inside play_steps' tail, in physical row order."""
        self.dataset_values = batch_dict

    def _amp_streams(self, idx):
        return None

    def _set_epoch_perm(self, perm):
        """Copy the mini-epoch permutation into persistent device storage (hipGraphs bind to slices of it)."""
        if getattr(self, '_perm_buf', None) is None:
            self._perm_buf = torch.empty(self.batch_size, dtype=torch.int32, device=self.ppo_device)
        self._perm_buf.copy_(perm)

    def _step(self, idx, new_z=None, slot=0):
        streams = self._amp_streams(idx)
        if self._use_rings:
            self.engine.use_slot(slot)           # this step's slot of the result ring (engine.set_result_slots)
        else:
            # without rings the step's launch program is keyed by its minibatch POSITION: the double-buffered inputs of the
            # un-chained prologue follow the position's parity (consecutive steps of a mini-epoch alternate; update() fences
            # the branch streams between mini-epochs)
            self.engine.use_parity(getattr(self, '_mb_pos', slot))
        if self.use_graph and new_z is None:
            return self._graph_step(idx, streams)
        return self.engine.step(self._ds, idx, self._remap, streams, new_z=new_z, fence=False)      # (update() fences per mini-epoch)

    def _graph_step(self, idx, streams):
        """Replay the optimisation step from a recorded launch sequence.  config['graph_capture']:
          True / 'program'  the library's own launch program (ase_hip_prog_*): the step's launches + fork / join points over
                            the engine's streams, replayed with 4-5 us of host work per launch (measured, scripts/lab/host_vs_gpu.py), branch -> stream mapping
                            fixed by us;
          'hipgraph'        a captured hipGraph (torch.cuda.CUDAGraph): the runtime chooses how its branches map to queues.
        Launch programs: ONE program for the whole step, data parallel included - the collectives are host-callback entries
        of the program (ase_hip_prog_host) at the positions the eager step issues them, so the discriminator bucket's
        all-reduce overlaps the policy branch's backward exactly as in eager mode.  Captured hipGraphs cannot hold a
        collective of another library: data parallel = three graphs (local statistics | forward-backward | optimizer) with
        the all-reduces between them.  One program (set) per minibatch position: the index tensors are
        slices of persistent per-mini-epoch buffers (permutation, composed demo / replay indices), so a replay needs no
        copies."""
        eng = self.engine
        # (result rings: the step's result slot is an argument of its launches - one program per optimisation step of the
        #  update instead of one per minibatch position)
        key = (int(idx.data_ptr()), eng.slot if self._use_rings else 0) + \
            (tuple((int(s[0].data_ptr()), int(s[1].data_ptr())) for s in streams) if streams else ())
        g = self._graphs.get(key)
        hipgraph = self.config.get('graph_capture') == 'hipgraph'
        single = (self.world_size == 1 and not eng.force_dist) or not hipgraph
        if g is None:
            eng.step(self._ds, idx, self._remap, streams, fence=False)   # this call's real step; also warms up lazies
            torch.cuda.synchronize()
            phases = [lambda: eng.step(self._ds, idx, self._remap, streams, fence=False)] if single else \
                [lambda: eng.phase_stats(self._ds, idx, self._remap, streams),
                 lambda: eng.phase_main(self._ds, idx, self._remap, streams),
                 lambda: eng.phase_apply(True)]
            graphs = []
            for fn in phases:
                if hipgraph:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        fn()
                    graphs.append(graph)
                else:
                    prog = self.backend.prog_create()
                    self.backend.prog_begin(prog)
                    try:
                        fn()
                    finally:
                        self.backend.prog_end(prog)
                    graphs.append(prog)
            # keep the captured index views alive: the replays read through their addresses
            self._graphs[key] = {'graphs': graphs, 'keep': (idx, streams), 'hipgraph': hipgraph}
            return eng.res
        run = (lambda x: x.replay()) if g['hipgraph'] else self.backend.prog_launch
        if len(g['graphs']) == 1:
            run(g['graphs'][0])
        else:
            run(g['graphs'][0])
            eng._allreduce_stats()
            run(g['graphs'][1])
            eng._allreduce_grads()
            run(g['graphs'][2])
        return eng.res

    def calc_gradients(self, input_dict):
        """Reference-compatible single step on an already gathered minibatch dict (learning/ase_agent.py:159)."""
        self.set_train()
        M = input_dict['obs'].shape[0]
        idx = torch.arange(M, dtype=torch.int32, device=self.ppo_device)
        streams = None
        if self.kind != 'ppo':
            streams = [(input_dict[k], idx, (0, 0)) for k in ('amp_obs', 'amp_obs_replay', 'amp_obs_demo')]
        ds = {k: v for k, v in input_dict.items() if v is not None}
        self.engine.step(ds, idx, (0, 0), streams, new_z=input_dict.get('_new_z'))
        self.train_result = self._collect_result()
        self._scaler_update()
        if self._snapshot_stream is not None:          # single-step callers read the result right away
            torch.cuda.current_stream().wait_stream(self._snapshot_stream)
            self._snapshot_stream = None

    def train_actor_critic(self, input_dict):
        self.calc_gradients(input_dict)
        return self.train_result

    def _collect_result(self):
        eng = self.engine
        # (a captured hipGraph replays on the main stream only: nothing would order its next replay behind a side-stream
        #  snapshot, so that mode snapshots in stream order)
        hipgraph = self.use_graph and self.config.get('graph_capture') == 'hipgraph'
        side = eng._side(1) if (eng.multi_stream and self._snapshot_aside and not hipgraph) else None
        if side is not None:
            # the two small snapshot copies (scalar vector, logit column) leave the main stream: they used to sit between
            # one step's last kernel and the next step's first (13 us per step).  They run on the discriminator's stream -
            # in front of the next step's discriminator work, which is what overwrites the logits; update() joins it.
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                r = dict(eng.results(snapshot=True))
            self._snapshot_stream = side
        else:
            r = dict(eng.results(snapshot=True))
        # (adaptive schedule: the rate the step was TAKEN with, recorded by the device before the schedule moved it)
        from .. import lib as L
        r['last_lr'] = eng.res.view(-1)[L.RES_LR].clone() if eng.adaptive_lr else self.last_lr
        r['lr_mul'] = 1.0
        return r

    def _pre_update(self, batch_dict):
        pass

    def _post_update(self, batch_dict):
        pass

    def train_epoch(self):
        play_time_start = time.time()
        with torch.no_grad():
            batch_dict = self.play_steps()
        play_time_end = time.time()
        update_time_start = time.time()
        train_info = self.update(batch_dict)
        update_time_end = time.time()
        train_info['play_time'] = play_time_end - play_time_start
        train_info['update_time'] = update_time_end - update_time_start
        train_info['total_time'] = update_time_end - play_time_start
        self._record_train_batch_info(batch_dict, train_info)
        return train_info

    # ------------------------------------------------------------------ the training loop (learning/common_agent.py:82-155)
    def _init_train(self):
        pass

    def train(self):
        """What rl_games' Runner.run calls: rollout + update until max_epochs, with the reference's bookkeeping (frame
        counters, reward / length meters, periodic checkpoints, per-epoch statistics sync under data parallel)."""
        self.init_tensors()
        self.last_mean_rewards = -100500
        total_time = 0
        self.frame = 0
        self.obs = self.env_reset() if hasattr(self.vec_env, 'reset') else None
        self.curr_frames = self.batch_size
        model_output_file = os.path.join(self.nn_dir, self.name)
        if self.multi_gpu:
            self._sync_initial_state()
        self._init_train()
        # config['manual_gc'] (off by default: it changes the process): the host is never more than one update ahead of the GPU
        # (every update ends in a small read-back), so a long host pause inside an update is GPU idle time.  With the switch on,
        # Python's cyclic collector runs between epochs only.  (The pauses actually measured on the benchmark boxes - 2 updates
        # in 20 at 1.2-2.3 x - were CPU-quota throttling from torch's OpenMP workers, see ase_amd.configure(cpu_threads=...).)
        manual_gc = bool(self.config.get('manual_gc', False))
        if manual_gc:
            import gc
            gc.collect()
            gc.freeze()
            gc.disable()
        while True:
            epoch_num = self.update_epoch()
            train_info = self.train_epoch()
            if manual_gc:
                gc.collect(1)
            sum_time = train_info['total_time']
            total_time += sum_time
            frame = self.frame
            if self.multi_gpu:
                self.sync_stats()
            if self.rank == 0:
                curr_frames = self.curr_frames
                self.frame += curr_frames
                if self.print_stats:
                    print(f"fps step: {curr_frames / max(train_info['play_time'], 1e-9):.1f} "
                          f"fps total: {curr_frames / max(sum_time, 1e-9):.1f}")
                self.writer.add_scalar('performance/total_fps', curr_frames / max(sum_time, 1e-9), frame)
                self.writer.add_scalar('performance/step_fps', curr_frames / max(train_info['play_time'], 1e-9), frame)
                self.writer.add_scalar('info/epochs', epoch_num, frame)
                self._log_train_info(train_info, frame)
                self.algo_observer.after_print_stats(frame, epoch_num, total_time)
                if self.game_rewards.current_size > 0:
                    mean_rewards = np.atleast_1d(self._get_mean_rewards())
                    mean_lengths = self.game_lengths.get_mean()
                    for i in range(self.value_size):
                        self.writer.add_scalar(f'rewards{i}/frame', mean_rewards[i], frame)
                        self.writer.add_scalar(f'rewards{i}/iter', mean_rewards[i], epoch_num)
                        self.writer.add_scalar(f'rewards{i}/time', mean_rewards[i], total_time)
                    self.writer.add_scalar('episode_lengths/frame', mean_lengths, frame)
                    self.writer.add_scalar('episode_lengths/iter', mean_lengths, epoch_num)
                if self.save_freq > 0 and epoch_num % self.save_freq == 0:
                    os.makedirs(self.nn_dir, exist_ok=True)
                    self.save(model_output_file)
                    if self._save_intermediate:
                        self.save(model_output_file + '_' + str(epoch_num).zfill(8))
            if epoch_num > self.max_epochs:
                if self.rank == 0:
                    os.makedirs(self.nn_dir, exist_ok=True)
                    self.save(model_output_file)
                    if self.print_stats:
                        print('MAX EPOCHS NUM!')
                if manual_gc:
                    gc.enable()
                return self.last_mean_rewards, epoch_num

    def _get_mean_rewards(self):
        return self.game_rewards.get_mean()

    def _record_train_batch_info(self, batch_dict, train_info):
        pass

    def _log_train_info(self, train_info, frame):
        w = self.writer
        w.add_scalar('performance/update_time', train_info['update_time'], frame)
        w.add_scalar('performance/play_time', train_info['play_time'], frame)
        w.add_scalar('losses/a_loss', _mean_list(train_info['actor_loss']).item(), frame)
        w.add_scalar('losses/c_loss', _mean_list(train_info['critic_loss']).item(), frame)
        w.add_scalar('losses/bounds_loss', _mean_list(train_info['b_loss']).item(), frame)
        w.add_scalar('losses/entropy', _mean_list(train_info['entropy']).item(), frame)
        w.add_scalar('info/last_lr', train_info['last_lr'][-1] * train_info['lr_mul'][-1], frame)
        w.add_scalar('info/lr_mul', train_info['lr_mul'][-1], frame)
        w.add_scalar('info/e_clip', self.e_clip * train_info['lr_mul'][-1], frame)
        w.add_scalar('info/clip_frac', _mean_list(train_info['actor_clip_frac']).item(), frame)
        w.add_scalar('info/kl', _mean_list(train_info['kl']).item(), frame)

    def update(self, batch_dict, perms=None, new_zs=None, max_steps=None):
        """Everything train_epoch does after the rollout (learning/amp_agent.py:194-262): the timed region of
        the benchmark together with the tail inside play_steps.  perms / new_zs: injected random draws
        (parity tests); otherwise drawn on the device.

        The update runs on the agent's own HIGH-priority stream (config['main_stream_priority'], default -1; 0 = the caller's
        stream): the step's critical path is the policy chain on the engine's main stream, and when its narrow launches
        (style MLP, heads, loss kernels) compete for CUs with the wide launches of the other branches the hardware serves the
        higher-priority queue first (MI355X, config 2 bf16: 66.4 -> 62.9 ms per update with the policy streams high and the
        discriminator's stream normal).  Ordered behind the caller's stream on entry, the caller's stream behind it on exit."""
        ms = self._main_stream
        if ms is None:
            return self._update(batch_dict, perms, new_zs, max_steps)
        cur = torch.cuda.current_stream(self.ppo_device)
        ms.wait_stream(cur)
        try:
            with torch.cuda.stream(ms):
                info = self._update(batch_dict, perms, new_zs, max_steps)
                # (what update() returns was allocated on the update's stream and is read on the caller's)
                bases = getattr(self, '_ring_bases', None) if self._use_rings else None
                if bases is not None:
                    for t in bases:
                        t.record_stream(cur)
                else:
                    for v in info.values() if info else ():
                        for t in v:
                            if torch.is_tensor(t) and t.is_cuda:
                                t.record_stream(cur)
        finally:
            # also when a step raised (OOM, AseHipError, a host callback): the caller's stream stays ordered behind whatever
            # the update's stream and its branch streams still hold - a checkpoint written next must not read half an update
            self.engine.fence_main_behind_sides(ms)
            cur.wait_stream(ms)
        return info

    def _update(self, batch_dict, perms=None, new_zs=None, max_steps=None):
        self._pre_update(batch_dict)
        self.set_train()
        self.curr_frames = batch_dict.pop('played_frames', self.batch_size)
        self.prepare_dataset(batch_dict)
        MB = self.minibatch_size
        # sharded mode: this rank's rows of every (global) minibatch; Horovod mode: the rank's own minibatches, whole
        R, rk = (self.world_size, self.rank) if self.dp_mode == 'shard' else (1, 0)
        m = MB // R
        train_info = None
        step = 0
        rings = self._use_rings
        for ep in range(self.mini_epochs_num):
            perm = self.dataset_perm if perms is None else perms[ep].to(self.ppo_device, torch.int32)
            self._set_epoch_perm(perm)
            perm = self._perm_buf                      # persistent storage: minibatch slices keep stable addresses
            # the branch streams read this mini-epoch's index buffers (and, in the first one, what _pre_update stored) without
            # waiting for the main stream inside a step (engine_opts xstep): order them here, once per mini-epoch
            self.engine.fence_side_streams()
            for i in range(self.num_minibatches):
                if max_steps is not None and step >= max_steps:
                    break
                mb_idx = perm[i * MB:(i + 1) * MB]
                self._mb_idx_full = mb_idx
                self._mb_pos = i
                idx = mb_idx[rk * m:(rk + 1) * m]
                self._step(idx, None if new_zs is None else new_zs[step].to(self.ppo_device)[rk * m:(rk + 1) * m], slot=step)
                if not rings:
                    cur = self._collect_result()
                    if train_info is None:
                        train_info = {k: [v] for k, v in cur.items()}
                    else:
                        for k, v in cur.items():
                            train_info[k].append(v)
                step += 1
            if perms is None:          # AMPDataset reshuffles after the last minibatch (learning/amp_datasets.py:24-30)
                self.dataset_perm = self._randperm(self.batch_size)
        if self._snapshot_stream is not None:
            torch.cuda.current_stream().wait_stream(self._snapshot_stream)
            self._snapshot_stream = None
        if rings:
            train_info = self._ring_results(step)
            self.train_result = {k: v[-1] for k, v in train_info.items()} if step else None
        self._post_update(batch_dict)
        self._lr_stale = self.engine.adaptive_lr      # (the host mirror `last_lr` is refreshed lazily: no read-back here)
        self._scaler_update()
        return train_info

    def _scaler_update(self):
        """mixed_precision with the dynamic loss scale: GradScaler.update() (learning/ase_agent.py:280,285,288) runs on the device
        behind every optimisation step (csrc/scaler.hip) and the recorded launch programs read the scale there - nothing to do
        between updates (round 5 moved the scale here, once per update, and dropped the programs)."""
        if self.engine.dyn_scale:
            self.engine.scaler_update()

    def _ring_results(self, n):
        """train_info of an update from the engine's result rings: ONE copy of the n result vectors (+ one of the n logit
        columns) on the main stream - every branch joined it at the end of its step - and per-step views of those copies
        under the reference's keys (learning/ase_agent.py:296-306, learning/common_agent.py:425-435)."""
        from .. import lib as L
        eng = self.engine
        R = eng.res_ring[:n].clone()
        self._ring_bases = [R]
        cols = {'entropy': L.RES_ENTROPY, 'kl': L.RES_KL, 'b_loss': L.RES_B_LOSS, 'actor_loss': L.RES_A_LOSS,
                'actor_clip_frac': L.RES_CLIP_FRAC, 'critic_loss': L.RES_C_LOSS, 'loss': L.RES_LOSS}
        if eng.has_disc:
            cols.update({'disc_loss': L.RES_DISC_LOSS, 'disc_grad_penalty': L.RES_DISC_GP, 'disc_logit_loss': L.RES_DISC_LOGIT_LOSS,
                         'disc_agent_acc': L.RES_DISC_AGENT_ACC, 'disc_demo_acc': L.RES_DISC_DEMO_ACC})
        if eng.has_enc:
            cols['enc_loss'] = L.RES_ENC_LOSS
            if eng.enc_gp:
                cols['enc_grad_penalty'] = L.RES_ENC_GP
        if eng.div_on:
            cols['amp_diversity_loss'] = L.RES_DIV_LOSS
        info = {k: [R[i, c] for i in range(n)] for k, c in cols.items()}
        if eng.has_disc:
            LG = eng.logit_ring[:n].clone()
            self._ring_bases.append(LG)
            a = 2 * eng.AMB
            info['disc_agent_logit'] = [LG[i, :a].view(-1, 1) for i in range(n)]
            info['disc_demo_logit'] = [LG[i, a:].view(-1, 1) for i in range(n)]
        # the learning rate every step was TAKEN with (train_result['last_lr'], learning/common_agent.py:430): adaptive schedule -
        # recorded on the device before the schedule moved it; constant - the host value
        info['last_lr'] = [R[i, L.RES_LR] for i in range(n)] if eng.adaptive_lr else [self._last_lr] * n
        info['lr_mul'] = [1.0] * n
        return info

    @property
    def last_lr(self):
        """Host mirror of the learning rate (checkpoints, logs).  With the adaptive schedule the rate lives on the device
        (ase_hip_finalize_scalars moves it every step): read back on demand, not at the end of every update."""
        if getattr(self, '_lr_stale', False) and getattr(self, 'engine', None) is not None:
            self._last_lr = float(self.engine.opt_state[1])
            self._lr_stale = False
        return self._last_lr

    @last_lr.setter
    def last_lr(self, v):
        self._last_lr = float(v)
        self._lr_stale = False


class AMPAgent(CommonAgent):
    kind = 'amp'

    def _load_config_params(self, config):
        super()._load_config_params(config)
        self._enable_eps_greedy = bool(config['enable_eps_greedy'])
        self._task_reward_w = config['task_reward_w']
        self._disc_reward_w = config['disc_reward_w']
        self._amp_observation_space = self.env_info['amp_observation_space']
        self._amp_batch_size = int(config['amp_batch_size'])
        self._amp_minibatch_size = int(config['amp_minibatch_size'])
        assert self._amp_minibatch_size <= self.minibatch_size               # learning/amp_agent.py:409
        self._disc_coef = config['disc_coef']
        self._disc_logit_reg = config['disc_logit_reg']
        self._disc_grad_penalty = config['disc_grad_penalty']
        self._disc_weight_decay = config['disc_weight_decay']
        self._disc_reward_scale = config['disc_reward_scale']
        self._normalize_amp_input = config.get('normalize_amp_input', True)
        self._amp_replay_keep_prob = config['amp_replay_keep_prob']

    def _build_net_config(self):
        c = super()._build_net_config()
        c['amp_input_shape'] = self._amp_observation_space.shape
        return c

    def init_tensors(self):
        super().init_tensors()
        H, N, dev = self.horizon_length, self.num_actors * self.num_agents, self.ppo_device
        A = self._amp_observation_space.shape[0]
        self.experience['amp_obs'] = torch.zeros(H, N, A, dtype=torch.float32, device=dev)
        self.experience['rand_action_mask'] = torch.zeros(H, N, dtype=torch.float32, device=dev)
        self._amp_obs_demo_buffer = ReplayBuffer(int(self.config['amp_obs_demo_buffer_size']), dev, self.backend, self._gen)
        self._amp_replay_buffer = ReplayBuffer(int(self.config['amp_replay_buffer_size']), dev, self.backend, self._gen)
        self.tensor_list += ['amp_obs', 'rand_action_mask']
        self._demo_ready = False
        self._build_rand_action_probs()

    def _build_rand_action_probs(self):
        """learning/amp_agent.py:424-435: env i acts stochastically with probability 1 - exp(10 (i / (N - 1) - 1))."""
        n = self.num_actors * self.num_agents
        env_ids = torch.arange(n, dtype=torch.float32, device=self.ppo_device)
        self._rand_action_probs = 1.0 - torch.exp(10 * (env_ids / (n - 1.0) - 1.0))
        self._rand_action_probs[0] = 1.0
        self._rand_action_probs[-1] = 0.0
        if not self._enable_eps_greedy:
            self._rand_action_probs[:] = 1.0

    def _init_train(self):
        super()._init_train()
        self._init_amp_demo_buf()

    def _act(self):
        return self.get_action_values(self.obs, self._rand_action_probs)

    def _rollout_extras(self, n, res_dict, infos):
        self.experience['amp_obs'][n].copy_(infos['amp_obs'])
        self.experience['rand_action_mask'][n].copy_(res_dict['rand_action_mask'])

    def _record_train_batch_info(self, batch_dict, train_info):
        super()._record_train_batch_info(batch_dict, train_info)
        train_info['disc_rewards'] = batch_dict['disc_rewards']

    def _log_train_info(self, train_info, frame):
        super()._log_train_info(train_info, frame)
        w = self.writer
        w.add_scalar('losses/disc_loss', _mean_list(train_info['disc_loss']).item(), frame)
        for k in ('disc_agent_acc', 'disc_demo_acc', 'disc_agent_logit', 'disc_demo_logit', 'disc_grad_penalty',
                  'disc_logit_loss'):
            w.add_scalar('info/' + k, _mean_list(train_info[k]).item(), frame)
        std, mean = torch.std_mean(train_info['disc_rewards'])
        w.add_scalar('info/disc_reward_mean', mean.item(), frame)
        w.add_scalar('info/disc_reward_std', std.item(), frame)

    def _experience_kwargs(self):
        return {'with_amp': True, 'with_latents': False}

    def get_stats_weights(self):
        state = super().get_stats_weights()
        if self._normalize_amp_input:
            state['amp_input_mean_std'] = self._rms_to_state_dict(self.engine.amp_state)
        return state

    def set_stats_weights(self, weights):
        super().set_stats_weights(weights)
        if self._normalize_amp_input:
            self._rms_from_state_dict(self.engine.amp_state, weights['amp_input_mean_std'])

    # ---- demo / replay plumbing (learning/amp_agent.py:498-533,579-593)
    def _fetch_amp_obs_demo(self, n):
        env = getattr(self.vec_env, 'env', self.vec_env)          # the reference reaches through vec_env.env
        return env.fetch_amp_obs_demo(n).to(self.ppo_device)

    def _init_amp_demo_buf(self):
        size = self._amp_obs_demo_buffer.get_buffer_size()
        for _ in range(int(np.ceil(size / self._amp_batch_size))):
            self._amp_obs_demo_buffer.store(self._fetch_amp_obs_demo(self._amp_batch_size))
        self._demo_ready = True

    def _update_amp_demos(self):
        self._amp_obs_demo_buffer.store(self._fetch_amp_obs_demo(self._amp_batch_size))

    def _store_replay_amp_obs(self, amp_obs_exp):
        buf = self._amp_replay_buffer
        size, total = buf.get_buffer_size(), buf.get_total_count()
        B = self.batch_size
        idx = None
        n = B
        if total > size and self.config.get('replay_keep_on_host', True):
            # (learning/amp_agent.py:579-593: once the ring is full every sample is kept with probability keep_prob.)  The mask
            # is drawn on the host: the NUMBER of kept rows moves the ring's head, and reading it back from the device ended
            # every update in a synchronisation (the GPU idle for the host's turn-around at every update boundary).
            keep = torch.bernoulli(torch.full((B,), float(self._amp_replay_keep_prob)), generator=self._host_gen) == 1.0
            idx_cpu = keep.nonzero(as_tuple=False).flatten().to(torch.int32)
            n = int(idx_cpu.numel())
            idx = self._upload_i32(idx_cpu, B)
        elif total > size:
            keep = torch.bernoulli(torch.full((B,), float(self._amp_replay_keep_prob), device=self.ppo_device),
                                   generator=self._gen) == 1.0
            idx = keep.nonzero(as_tuple=False).flatten().to(torch.int32)
            n = int(idx.numel())
        if n > size:
            sel = torch.randperm(n, device=self.ppo_device, generator=self._gen)[:size]
            idx = sel.to(torch.int32) if idx is None else idx[sel.long()]
            n = size
        buf.store(amp_obs_exp.view(B, -1), n=n, idx=idx, remap=self._remap)

    def _pre_update(self, batch_dict):
        if not self._demo_ready:
            self._init_amp_demo_buf()
        self._update_amp_demos()
        B = self.batch_size
        self._demo_idx = self._amp_obs_demo_buffer.sample_indices(B)
        if self._amp_replay_buffer.get_total_count() == 0:
            self._replay_idx = None                   # amp_obs_replay = amp_obs (learning/amp_agent.py:199-200)
        else:
            self._replay_idx = self._amp_replay_buffer.sample_indices(B)

    def _post_update(self, batch_dict):
        self._store_replay_amp_obs(self.experience['amp_obs'])

    def _set_epoch_perm(self, perm):
        """+ demo / replay ring indices composed with the permutation, once per mini-epoch:
        amp_obs_demo[perm[j]] = demo_ring[demo_idx[perm[j]]] (learning/amp_agent.py:196, learning/amp_datasets.py:21-22)."""
        super()._set_epoch_perm(perm)
        if getattr(self, '_demo_comp', None) is None:
            self._demo_comp = torch.empty_like(self._perm_buf)
            self._replay_comp = torch.empty_like(self._perm_buf)
        pl = self._perm_buf.long()
        self._demo_comp.copy_(self._demo_idx[pl])
        if self._replay_idx is not None:
            self._replay_comp.copy_(self._replay_idx[pl])

    def _amp_streams(self, idx):
        """agent / replay / demo rows of this minibatch as (source, index, remap) — the first amp_minibatch rows
        of the minibatch (learning/ase_agent.py:172-181), this rank's share of them."""
        R, rk = (self.world_size, self.rank) if self.dp_mode == 'shard' else (1, 0)
        amb = self._amp_minibatch_size
        a = amb // R
        lo = self._mb_pos * self.minibatch_size + rk * a          # position of this rank's amp rows in the permutation
        rows = self._perm_buf[lo:lo + a]
        agent = (self._ds['amp_obs'], rows, self._remap)
        demo = (self._amp_obs_demo_buffer.data, self._demo_comp[lo:lo + a], (0, 0))
        if self._replay_idx is None:
            replay = (self._ds['amp_obs'], rows, self._remap)
        else:
            replay = (self._amp_replay_buffer.data, self._replay_comp[lo:lo + a], (0, 0))
        return [agent, replay, demo]

    def _calc_disc_rewards(self, amp_obs):
        HD, _ = self.engine.amp_heads(amp_obs.view(-1, amp_obs.shape[-1]))
        r = torch.empty(HD.shape[0], 1, dtype=torch.float32, device=self.ppo_device)
        self.backend.disc_reward(HD, r, HD.shape[0], self._disc_reward_scale)
        return r.view(*amp_obs.shape[:-1], 1)

    def _calc_amp_rewards(self, amp_obs):
        return {'disc_rewards': self._calc_disc_rewards(amp_obs)}


class ASEAgent(AMPAgent):
    kind = 'ase'

    def _load_config_params(self, config):
        super()._load_config_params(config)
        self._latent_dim = config['latent_dim']
        self._latent_steps_min = config.get('latent_steps_min', np.inf)
        self._latent_steps_max = config.get('latent_steps_max', np.inf)
        self._amp_diversity_bonus = config['amp_diversity_bonus']
        self._amp_diversity_tar = config['amp_diversity_tar']
        self._enc_coef = config['enc_coef']
        self._enc_weight_decay = config['enc_weight_decay']
        self._enc_reward_scale = config['enc_reward_scale']
        self._enc_grad_penalty = config['enc_grad_penalty']
        self._enc_reward_w = config['enc_reward_w']

    def _build_net_config(self):
        c = super()._build_net_config()
        c['ase_latent_shape'] = (self._latent_dim,)
        return c

    def init_tensors(self):
        super().init_tensors()
        H, N, dev = self.horizon_length, self.num_actors * self.num_agents, self.ppo_device
        self.experience['ase_latents'] = torch.zeros(H, N, self._latent_dim, dtype=torch.float32, device=dev)
        self.tensor_list += ['ase_latents']
        self._ase_latents = torch.zeros(N, self._latent_dim, dtype=torch.float32, device=dev)      # learning/ase_agent.py:24-27
        self._latent_reset_steps = torch.zeros(N, dtype=torch.int32, device=dev)

    # ---- per-environment latents (learning/ase_agent.py:310-379)
    def _progress_buf(self):
        env = getattr(self.vec_env, 'env', self.vec_env)
        return env.task.progress_buf.to(self.ppo_device)

    def env_reset(self, env_ids=None):
        obs = super().env_reset(env_ids)
        if env_ids is None:
            env_ids = torch.arange(self.num_actors * self.num_agents, dtype=torch.long, device=self.ppo_device)
        if len(env_ids) > 0:
            env_ids = torch.as_tensor(env_ids, dtype=torch.long, device=self.ppo_device)
            self._reset_latents(env_ids)
            self._reset_latent_step_count(env_ids)
        return obs

    def _rand_steps(self, n):
        return torch.randint(int(self._latent_steps_min), int(self._latent_steps_max), (n,), device=self.ppo_device,
                             generator=self._gen, dtype=torch.int32)

    def _reset_latent_step_count(self, env_ids):
        self._latent_reset_steps[env_ids] = self._rand_steps(len(env_ids))

    def _sample_latents(self, n):
        return self.model.a2c_network.sample_latents(n)

    def _reset_latents(self, env_ids):
        self._ase_latents[env_ids] = self._sample_latents(len(env_ids))

    def _update_latents(self):
        new_latent_envs = self._latent_reset_steps <= self._progress_buf()
        if bool(torch.any(new_latent_envs)):
            ids = new_latent_envs.nonzero(as_tuple=False).flatten()
            self._reset_latents(ids)
            self._latent_reset_steps[ids] += self._rand_steps(len(ids))

    def _before_act(self):
        self._update_latents()

    def _act(self):
        return self.get_action_values(self.obs, self._ase_latents, self._rand_action_probs)

    def _next_values(self):
        return self._eval_critic(self.obs, self._ase_latents)

    def _rollout_extras(self, n, res_dict, infos):
        super()._rollout_extras(n, res_dict, infos)
        self.experience['ase_latents'][n].copy_(self._ase_latents)

    def _record_train_batch_info(self, batch_dict, train_info):
        super()._record_train_batch_info(batch_dict, train_info)
        train_info['enc_rewards'] = batch_dict['enc_rewards']

    def _log_train_info(self, train_info, frame):
        super()._log_train_info(train_info, frame)
        w = self.writer
        w.add_scalar('losses/enc_loss', _mean_list(train_info['enc_loss']).item(), frame)
        if self._amp_diversity_bonus != 0:
            w.add_scalar('losses/amp_diversity_loss', _mean_list(train_info['amp_diversity_loss']).item(), frame)
        std, mean = torch.std_mean(train_info['enc_rewards'])
        w.add_scalar('info/enc_reward_mean', mean.item(), frame)
        w.add_scalar('info/enc_reward_std', std.item(), frame)
        if self._enc_grad_penalty != 0:                                  # learning/ase_agent.py:509-510
            w.add_scalar('info/enc_grad_penalty', _mean_list(train_info['enc_grad_penalty']).item(), frame)

    def _experience_kwargs(self):
        return {'with_amp': True, 'with_latents': True}

    def _calc_amp_rewards(self, amp_obs, ase_latents):
        HD, enc = self.engine.amp_heads(amp_obs.view(-1, amp_obs.shape[-1]))
        n = HD.shape[0]
        rd = torch.empty(n, 1, dtype=torch.float32, device=self.ppo_device)
        re = torch.empty(n, 1, dtype=torch.float32, device=self.ppo_device)
        self.backend.disc_reward(HD, rd, n, self._disc_reward_scale)
        self.backend.enc_reward(enc, ase_latents.view(n, -1), re, n, self._latent_dim, self._enc_reward_scale)
        shp = tuple(amp_obs.shape[:-1]) + (1,)
        return {'disc_rewards': rd.view(shp), 'enc_rewards': re.view(shp)}


class HRLAgent(CommonAgent):
    """learning/hrl_agent.py:26-268: a trainable high-level policy (plain PPO on the tanh-mu A2C net,
    learning/hrl_network_builder.py:26-29) whose 64-dim action is the latent of a FROZEN ASE low-level controller.
    One high-level step = `llc_steps` simulator steps: z = normalize(action) -> LLC actor (HIP inference path, deterministic
    mu) -> env.step, with the LLC discriminator's reward averaged over the inner steps (hrl_agent.py:45-82,231-249).

    config['llc_config']: path of the LLC's yaml or the parsed dict; config['llc_checkpoint']: path of a checkpoint written
    by ASEAgent.save (or the reference) or the weights dict itself."""
    kind = 'ppo'

    def __init__(self, base_name, config):
        llc = config['llc_config']
        if not isinstance(llc, dict):
            import yaml
            with open(os.path.join(os.getcwd(), llc), 'r') as f:
                llc = yaml.load(f, Loader=yaml.SafeLoader)
        self._llc_params = llc['params']
        self._latent_dim = self._llc_params['config']['latent_dim']
        super().__init__(base_name, config)
        env = getattr(self.vec_env, 'env', self.vec_env)
        self._task_size = env.task.get_task_obs_size()
        self._llc_steps = config['llc_steps']
        llc_checkpoint = config['llc_checkpoint']
        assert llc_checkpoint != ""                                           # learning/hrl_agent.py:40
        self._build_llc(self._llc_params, llc_checkpoint)

    def _load_config_params(self, config):
        super()._load_config_params(config)
        self._task_reward_w = config['task_reward_w']
        self._disc_reward_w = config['disc_reward_w']

    def _setup_action_space(self):
        super()._setup_action_space()             # low / high of the ENVIRONMENT's action space: what the LLC's mu is scaled to
        self.actions_num = self._latent_dim       # the high-level policy acts in latent space (hrl_agent.py:181-184)

    def init_tensors(self):
        super().init_tensors()
        H, N, dev = self.horizon_length, self.num_actors * self.num_agents, self.ppo_device
        self.experience['disc_rewards'] = torch.zeros(H, N, 1, dtype=torch.float32, device=dev)
        self.tensor_list += ['disc_rewards']

    # ---- the frozen low-level controller
    def _build_llc(self, config_params, checkpoint):
        from . import models
        from .network_builder import ASEBuilder
        import copy
        import types
        builder = ASEBuilder()
        builder.load(config_params['network'])
        obs_size = self.obs_shape[0] - self._task_size
        env_info = dict(self.env_info)
        env_info['observation_space'] = types.SimpleNamespace(shape=(obs_size,))
        env = getattr(self.vec_env, 'env', self.vec_env)
        if 'amp_observation_space' not in env_info:
            env_info['amp_observation_space'] = env.amp_observation_space
        cfg = copy.copy(config_params['config'])
        cfg.update(network=models.ModelASEContinuous(builder), num_actors=self.num_actors, env_info=env_info,
                   device=self.ppo_device, vec_env=None, features={'observer': self.algo_observer},
                   precision=self.config.get('precision', 'bf16'), seed=self.seed)
        if self.config.get('backend') is not None:
            cfg['backend'] = self.config['backend']
        self._llc_agent = ASEAgent('llc', cfg)
        if isinstance(checkpoint, dict):
            self._llc_agent.set_full_state_weights(checkpoint)
        else:
            self._llc_agent.restore(checkpoint)
        self._llc_agent.set_eval()

    def _extract_llc_obs(self, obs):
        return obs[..., :obs.shape[-1] - self._task_size]

    def _compute_llc_action(self, obs, actions):
        """hrl_agent.py:231-241: normalised LLC observation, z = normalize(high-level action), deterministic LLC mu,
        scaled to the environment's action range (ASEAgent.preprocess_actions)."""
        llc_obs = self._extract_llc_obs(obs).contiguous()
        e = self._llc_agent.engine
        z = e._scr('hrl_z', actions.shape[0], self._latent_dim, torch.float32)
        self.backend.normalize_rows(actions, z, actions.shape[0], self._latent_dim)
        mu = e.policy_forward(llc_obs, z, want=('mu',))['mu']
        return self._llc_agent.preprocess_actions(mu)

    def _calc_disc_reward(self, amp_obs):
        return self._llc_agent._calc_disc_rewards(amp_obs)

    def preprocess_actions(self, actions):
        return torch.clamp(actions, -1.0, 1.0)                                # hrl_agent.py:90-94

    def env_step(self, actions):
        actions = self.preprocess_actions(actions)
        obs = self.obs['obs']
        rewards = disc_rewards = done_count = terminate_count = 0.0
        for t in range(self._llc_steps):
            llc_actions = self._compute_llc_action(obs, actions)
            obs, curr_rewards, curr_dones, infos = self.vec_env.step(llc_actions)
            obs = obs['obs'] if isinstance(obs, dict) else obs
            obs = obs.to(self.ppo_device)
            rewards = rewards + curr_rewards.to(self.ppo_device)
            done_count = done_count + curr_dones.to(self.ppo_device).float()
            terminate_count = terminate_count + infos['terminate'].to(self.ppo_device).float()
            disc_rewards = disc_rewards + self._calc_disc_reward(infos['amp_obs'].to(self.ppo_device))
        rewards = rewards / self._llc_steps
        disc_rewards = disc_rewards / self._llc_steps
        dones = (done_count > 0).to(done_count.dtype)
        infos['terminate'] = (terminate_count > 0).to(terminate_count.dtype)
        infos['disc_rewards'] = disc_rewards
        if self.value_size == 1:
            rewards = rewards.unsqueeze(1)
        return self.obs_to_tensors(obs), rewards, dones, infos

    def _rollout_extras(self, n, res_dict, infos):
        self.experience['disc_rewards'][n].copy_(infos['disc_rewards'].view(-1, 1))

    def _get_mean_rewards(self):
        return super()._get_mean_rewards() * self._llc_steps

    def _record_train_batch_info(self, batch_dict, train_info):
        super()._record_train_batch_info(batch_dict, train_info)
        train_info['disc_rewards'] = batch_dict['disc_rewards']

    def _log_train_info(self, train_info, frame):
        super()._log_train_info(train_info, frame)
        std, mean = torch.std_mean(train_info['disc_rewards'])
        self.writer.add_scalar('info/disc_reward_mean', mean.item(), frame)
        self.writer.add_scalar('info/disc_reward_std', std.item(), frame)
