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
import time

import numpy as np
import torch

from .. import lib as L
from ..engine import UpdateEngine
from ..inference import InferenceEngine
from .replay_buffer import ReplayBuffer


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
        self.truncate_grads = config.get('truncate_grads', False)
        assert not self.truncate_grads, "truncate_grads: SURVEY §8(f) row N4 (off in every reference config)"
        assert not config.get('mixed_precision', False), "use config['precision'] = 'bf16' | 'f32'"
        assert config.get('lr_schedule', 'constant') in ('constant', None)
        self.multi_gpu = config.get('multi_gpu', False)
        self.world_size, self.rank = config.get('world_size', 1), config.get('rank', 0)
        self.last_lr = float(config['learning_rate'])
        self.epoch_num = 0
        self.frame = 0
        self.last_mean_rewards = -100500
        self.is_rnn = False
        self.obs_shape = self.env_info['observation_space'].shape
        self.actions_num = self.env_info['action_space'].shape[0]
        self.value_size = self.env_info.get('value_size', 1)
        self.vec_env = config.get('vec_env', None)
        self._load_config_params(config)

        self.network = config['network']
        self.model = self.network.build(self._build_net_config())
        self.model.to(self.ppo_device)
        # precision: 'bf16' (bf16 storage + MFMA, throughput mode) | 'f32' (exact f32 MFMA) |
        #            'bf16x3' (f32 storage, every product as three bf16 MFMAs on a hi/lo split: f32-grade results)
        precision = config.get('precision', 'bf16')
        dtype = {'bf16': torch.bfloat16, 'f32': torch.float32, 'bf16x3': torch.float32}[precision]
        backend = config.get('backend', None)
        if backend is None:
            from ..backend import HipBackend
            backend = HipBackend(self.ppo_device, x3=(precision == 'bf16x3'))
        self.backend = backend
        self.engine = UpdateEngine(self.kind, self.model.a2c_network, config, backend, minibatch=self.minibatch_size,
                                   amp_minibatch=getattr(self, '_amp_minibatch_size', 0), dtype=dtype,
                                   world_size=self.world_size, rank=self.rank)
        self.model.a2c_network.infer = InferenceEngine(self.model.a2c_network, self.engine)
        self.use_graph = bool(config.get('graph_capture', False))
        self._graphs = {}
        self._train_mode = True
        self.train_result = None
        self.dataset_perm = torch.randperm(self.batch_size, device=self.ppo_device).to(torch.int32)
        self.init_tensors()

    # ------------------------------------------------------------------ config
    def _load_config_params(self, config):
        pass

    def _build_net_config(self):
        return {'actions_num': self.actions_num, 'input_shape': self.obs_shape,
                'num_seqs': self.num_actors * self.num_agents, 'value_size': self.value_size,
                'device': self.ppo_device}

    # ------------------------------------------------------------------ buffers
    def init_tensors(self):
        H, N, dev = self.horizon_length, self.num_actors * self.num_agents, self.ppo_device
        f32 = dict(dtype=torch.float32, device=dev)
        A = self.actions_num
        self.experience = {            # rl_games ExperienceBuffer.tensor_dict layout (time-major)
            'obses': torch.zeros(H, N, *self.obs_shape, **f32), 'rewards': torch.zeros(H, N, 1, **f32),
            'values': torch.zeros(H, N, 1, **f32), 'neglogpacs': torch.zeros(H, N, **f32),
            'dones': torch.zeros(H, N, dtype=torch.uint8, device=dev), 'actions': torch.zeros(H, N, A, **f32),
            'mus': torch.zeros(H, N, A, **f32), 'sigmas': torch.zeros(H, N, A, **f32),
            'next_obses': torch.zeros(H, N, *self.obs_shape, **f32), 'next_values': torch.zeros(H, N, 1, **f32)}
        self.tensor_list = ['actions', 'neglogpacs', 'values', 'mus', 'sigmas', 'obses', 'states', 'dones', 'next_obses']

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

    def get_full_state_weights(self):
        state = self.get_weights()
        state['epoch'] = self.epoch_num
        state['optimizer'] = self._optimizer_state_dict()
        state['frame'] = self.frame
        state['last_mean_rewards'] = self.last_mean_rewards
        state['env_state'] = None
        return state

    def set_full_state_weights(self, weights):
        self.set_weights(weights)
        self.epoch_num = weights['epoch']
        self._load_optimizer_state_dict(weights['optimizer'])
        self.frame = weights.get('frame', 0)
        self.last_mean_rewards = weights.get('last_mean_rewards', -100500)

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

    def get_action_values(self, obs_dict, *extra):
        mu, sigma, value = self._policy(obs_dict['obs'], *extra[:1] if self.kind == 'ase' else ())
        action = mu + sigma * torch.randn_like(mu)
        logstd = torch.log(sigma)
        nlp = 0.5 * (((action - mu) / sigma) ** 2).sum(-1) + 0.5 * np.log(2 * np.pi) * mu.shape[-1] + logstd.sum(-1)
        return {'neglogpacs': nlp, 'values': value, 'actions': action, 'mus': mu, 'sigmas': sigma, 'rnn_states': None}

    def _eval_critic(self, obs_dict, *extra):
        return self.engine.policy_forward(obs_dict['obs'], *extra[:1], want=('value',))['value']

    def play_steps(self):
        """Fill the experience buffer from the (synthetic) environment, then run the reference's tail."""
        self.set_eval()
        exp = self.vec_env.experience(self._cpu_policy(), **self._experience_kwargs())
        for k, v in exp.items():
            if k in self.experience:
                self.experience[k].copy_(v.to(self.ppo_device))
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

    def _step(self, idx, new_z=None):
        streams = self._amp_streams(idx)
        if self.use_graph and new_z is None:
            return self._graph_step(idx, streams)
        return self.engine.step(self._ds, idx, self._remap, streams, new_z=new_z)

    def _graph_step(self, idx, streams):
        """Replay the optimisation step from captured hipGraphs.  Single GPU: one graph for the whole step.
        Data parallel: three graphs (local statistics | forward-backward | optimizer) with the two RCCL all-reduces
        issued between them.  One graph (set) per minibatch position: the index tensors are slices of persistent
        per-mini-epoch buffers (permutation, composed demo / replay indices), so a replay needs no copies."""
        eng = self.engine
        key = (int(idx.data_ptr()),) + (tuple((int(s[0].data_ptr()), int(s[1].data_ptr())) for s in streams) if streams else ())
        g = self._graphs.get(key)
        if g is None:
            eng.step(self._ds, idx, self._remap, streams)                # this call's real step; also warms up lazies
            torch.cuda.synchronize()
            if self.world_size == 1 and not eng.force_dist:
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    eng.step(self._ds, idx, self._remap, streams)
                graphs = [graph]
            else:
                ga, gb, gc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga):
                    eng.phase_stats(self._ds, idx, self._remap, streams)
                with torch.cuda.graph(gb):
                    eng.phase_main(self._ds, idx, self._remap, streams)
                with torch.cuda.graph(gc):
                    eng.phase_apply(True)
                graphs = [ga, gb, gc]
            # keep the captured index views alive: the graphs read through their addresses on every replay
            self._graphs[key] = {'graphs': graphs, 'keep': (idx, streams)}
            return eng.res
        if len(g['graphs']) == 1:
            g['graphs'][0].replay()
        else:
            g['graphs'][0].replay()
            eng._allreduce_stats()
            g['graphs'][1].replay()
            eng._allreduce_grads()
            g['graphs'][2].replay()
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

    def train_actor_critic(self, input_dict):
        self.calc_gradients(input_dict)
        return self.train_result

    def _collect_result(self):
        r = dict(self.engine.results(snapshot=True))
        r['last_lr'] = self.last_lr
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
        return train_info

    def update(self, batch_dict, perms=None, new_zs=None, max_steps=None):
        """Everything train_epoch does after the rollout (learning/amp_agent.py:194-262): the timed region of
        the benchmark together with the tail inside play_steps.  perms / new_zs: injected random draws
        (parity tests); otherwise drawn on the device."""
        self._pre_update(batch_dict)
        self.set_train()
        self.curr_frames = batch_dict.pop('played_frames', self.batch_size)
        self.prepare_dataset(batch_dict)
        MB, R, rk = self.minibatch_size, self.world_size, self.rank
        m = MB // R
        train_info = None
        step = 0
        for ep in range(self.mini_epochs_num):
            perm = self.dataset_perm if perms is None else perms[ep].to(self.ppo_device, torch.int32)
            self._set_epoch_perm(perm)
            perm = self._perm_buf                      # persistent storage: minibatch slices keep stable addresses
            for i in range(self.num_minibatches):
                if max_steps is not None and step >= max_steps:
                    break
                mb_idx = perm[i * MB:(i + 1) * MB]
                self._mb_idx_full = mb_idx
                self._mb_pos = i
                idx = mb_idx[rk * m:(rk + 1) * m]
                self._step(idx, None if new_zs is None else new_zs[step].to(self.ppo_device)[rk * m:(rk + 1) * m])
                cur = self._collect_result()
                if train_info is None:
                    train_info = {k: [v] for k, v in cur.items()}
                else:
                    for k, v in cur.items():
                        train_info[k].append(v)
                step += 1
            if perms is None:          # AMPDataset reshuffles after the last minibatch (learning/amp_datasets.py:24-30)
                self.dataset_perm = torch.randperm(self.batch_size, device=self.ppo_device).to(torch.int32)
        self._post_update(batch_dict)
        return train_info


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
        self._amp_obs_demo_buffer = ReplayBuffer(int(self.config['amp_obs_demo_buffer_size']), dev, self.backend)
        self._amp_replay_buffer = ReplayBuffer(int(self.config['amp_replay_buffer_size']), dev, self.backend)
        self.tensor_list += ['amp_obs', 'rand_action_mask']
        self._demo_ready = False

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
        return self.vec_env.fetch_amp_obs_demo(n).to(self.ppo_device)

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
        if total > size:
            keep = torch.bernoulli(torch.full((B,), float(self._amp_replay_keep_prob), device=self.ppo_device)) == 1.0
            idx = keep.nonzero(as_tuple=False).flatten().to(torch.int32)
            n = int(idx.numel())
        if n > size:
            sel = torch.randperm(n, device=self.ppo_device)[:size]
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
        R, rk = self.world_size, self.rank
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
    """High-level policy update of the HRL agent = CommonAgent's plain PPO step on the tanh-mu A2C net
    (learning/hrl_agent.py:26, learning/hrl_network_builder.py:26-29).  The frozen low-level controller and
    its 5-step env_step (hrl_agent.py:45-82) belong to the rollout side."""
    kind = 'ppo'
