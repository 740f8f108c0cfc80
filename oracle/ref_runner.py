"""Run the reference's OWN learning code (``/root/reference/ase/learning/*.py``, imported
unmodified) on CPU through the restated rl_games shim.  TEST INFRASTRUCTURE ONLY.

Only usable in the authoring container, where ``/root/reference`` is mounted.  It exists to
(a) generate the golden vectors under ``tests/golden/`` (``oracle/make_golden.py``) and
(b) pin ``oracle/restated.py`` (the travel-capable restatement) against the real code.

Nothing under ``ase_amd/`` may import this module.

Agents are built by ``object.__new__`` + attribute injection instead of rl_games'
``A2CBase.__init__`` (which would need Isaac Gym, a vec-env registry and tensorboard): every
attribute set below is one that ``A2CBase.__init__`` derives from the yaml ``params.config``
block; every *method* that then runs — ``_load_config_params``,
``_build_net_config``, ``_build_amp_buffers``, ``prepare_dataset``, ``calc_gradients``,
``discount_values``, ``_calc_advs``, ``_calc_amp_rewards``, ``train_epoch`` … — is the reference's.
"""
import os
import sys
import types

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("ASE_REFERENCE_ROOT", "/root/reference")
_SHIM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rl_games_shim")

_loaded = None


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "ase", "learning"))


def load_reference():
    """Import the reference's learning package (unmodified) and return its modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference tree not mounted at %s" % REFERENCE_ROOT)
    for p in (_SHIM_DIR, os.path.join(REFERENCE_ROOT, "ase")):
        if p not in sys.path:
            sys.path.insert(0, p)
    # ase/utils/torch_utils.py needs isaacgym's quaternion helpers; the agents import it
    # (ase_agent.py:9) but never call it, so an empty module is exact for the hot path.
    if "utils" not in sys.modules:
        u = types.ModuleType("utils")
        tu = types.ModuleType("utils.torch_utils")
        u.torch_utils = tu
        sys.modules["utils"] = u
        sys.modules["utils.torch_utils"] = tu
    import contextlib
    import io
    sys.dont_write_bytecode = True          # /root/reference is read-only material: no __pycache__ next to its sources
    with contextlib.redirect_stdout(io.StringIO()):
        from learning import (amp_agent, amp_datasets, amp_models, amp_network_builder, ase_agent,
                              ase_models, ase_network_builder, common_agent, hrl_models,
                              hrl_network_builder, replay_buffer)
    import torch.nn as nn
    # ase_agent.py:277,283 and amp_agent.py use `nn` (truncate_grads branch); amp_agent imports it,
    # ase_agent does not (latent NameError in the reference, dead under the default config).
    ase_agent.nn = nn
    _loaded = types.SimpleNamespace(
        amp_agent=amp_agent, amp_datasets=amp_datasets, amp_models=amp_models,
        amp_network_builder=amp_network_builder, ase_agent=ase_agent, ase_models=ase_models,
        ase_network_builder=ase_network_builder, common_agent=common_agent,
        hrl_models=hrl_models, hrl_network_builder=hrl_network_builder,
        replay_buffer=replay_buffer)
    return _loaded


class _Quiet:
    def __enter__(self):
        import contextlib
        import io
        self._cm = contextlib.redirect_stdout(io.StringIO())
        self._cm.__enter__()

    def __exit__(self, *a):
        self._cm.__exit__(*a)


class _Observer:
    def after_init(self, algo):
        pass

    def after_steps(self):
        pass

    def process_infos(self, infos, done_indices):
        pass

    def after_print_stats(self, *a):
        pass


class _ExperienceBuffer:
    """rl_games.common.experience.ExperienceBuffer shapes/dtypes."""

    def __init__(self, horizon, num_actors, obs_shape, actions_num, value_size, device):
        self.obs_base_shape = (horizon, num_actors)
        b = self.obs_base_shape
        f32 = dict(dtype=torch.float32, device=device)
        self.tensor_dict = {
            'obses': torch.zeros(b + tuple(obs_shape), **f32),
            'rewards': torch.zeros(b + (value_size,), **f32),
            'values': torch.zeros(b + (value_size,), **f32),
            'neglogpacs': torch.zeros(b, **f32),
            'dones': torch.zeros(b, dtype=torch.uint8, device=device),
            'actions': torch.zeros(b + (actions_num,), **f32),
            'mus': torch.zeros(b + (actions_num,), **f32),
            'sigmas': torch.zeros(b + (actions_num,), **f32),
        }

    def update_data(self, name, index, val):
        self.tensor_dict[name][index, :] = val

    def get_transformed_list(self, transform_op, tensor_list):
        res = {}
        for k in tensor_list:
            v = self.tensor_dict.get(k)
            if v is None:
                continue
            res[k] = transform_op(v)
        return res


def build_ref_agent(kind, net_params, config, *, num_envs, obs_size, act_size, amp_obs_size=None,
                    device='cpu', demo_fetch=None, seed=0):
    """kind in {'amp','ase','ppo'}: returns a reference agent ready for the update path.

    net_params: yaml ``params.network`` subtree; config: yaml ``params.config`` subtree."""
    ref = load_reference()
    from gym import spaces
    from rl_games.algos_torch.running_mean_std import RunningMeanStd
    from rl_games.common import schedulers

    torch.manual_seed(seed)
    if kind == 'ase':
        agent_cls, builder, model_cls = ref.ase_agent.ASEAgent, ref.ase_network_builder.ASEBuilder(), ref.ase_models.ModelASEContinuous
    elif kind == 'amp':
        agent_cls, builder, model_cls = ref.amp_agent.AMPAgent, ref.amp_network_builder.AMPBuilder(), ref.amp_models.ModelAMPContinuous
    elif kind == 'ppo':  # CommonAgent with the HRL high-level net (plain PPO update, common_agent.py:353)
        agent_cls, builder, model_cls = ref.common_agent.CommonAgent, ref.hrl_network_builder.HRLBuilder(), ref.hrl_models.ModelHRLContinuous
    else:
        raise ValueError(kind)
    builder.load(net_params)

    A = object.__new__(agent_cls)
    A.config = config
    A.env_info = {
        'observation_space': spaces.Box(-np.inf, np.inf, shape=(obs_size,)),
        'action_space': spaces.Box(-1.0, 1.0, shape=(act_size,)),
    }
    if amp_obs_size is not None:
        A.env_info['amp_observation_space'] = spaces.Box(-np.inf, np.inf, shape=(amp_obs_size,))
    A.ppo_device = device
    A.rank = 0
    A.multi_gpu = False
    A.num_actors = num_envs
    A.num_agents = 1
    A.value_size = 1
    A.obs_shape = (obs_size,)
    A.observation_space = A.env_info['observation_space']
    A.weight_decay = config.get('weight_decay', 0.0)
    A.use_action_masks = False
    A.is_train = True
    A.central_value_config = None
    A.has_central_value = False
    A.truncate_grads = config.get('truncate_grads', False)
    A.ppo = config['ppo']
    A.max_epochs = config.get('max_epochs', 1e6)
    A.is_adaptive_lr = config['lr_schedule'] == 'adaptive'
    assert config['lr_schedule'] in ('constant', None)
    A.scheduler = schedulers.IdentityScheduler()
    A.schedule_type = config.get('schedule_type', 'legacy')
    A.e_clip = config['e_clip']
    A.clip_value = config['clip_value']
    A.horizon_length = config['horizon_length']
    A.seq_len = config.get('seq_length', 4)
    A.normalize_advantage = config['normalize_advantage']
    A.normalize_input = config['normalize_input']
    A.normalize_value = config.get('normalize_value', False)
    A.critic_coef = config['critic_coef']
    A.grad_norm = config['grad_norm']
    A.gamma = config['gamma']
    A.tau = config['tau']
    A.minibatch_size = config['minibatch_size']
    A.mini_epochs_num = config['mini_epochs']
    A.mixed_precision = config.get('mixed_precision', False)
    A.entropy_coef = config['entropy_coef']
    A.batch_size = A.horizon_length * A.num_actors * A.num_agents
    A.batch_size_envs = A.horizon_length * A.num_actors
    assert A.batch_size % A.minibatch_size == 0
    A.num_minibatches = A.batch_size // A.minibatch_size
    A.is_rnn = False
    A.rnn_states = None
    A.epoch_num = 0
    A.frame = 0
    A.last_lr = config['learning_rate']
    A.scaler = torch.amp.GradScaler('cpu', enabled=False)
    A.algo_observer = _Observer()
    A.writer = None
    A.is_tensor_obses = True
    A.hvd = None
    if A.normalize_value:
        A.value_mean_std = RunningMeanStd((1,)).to(device)

    # --- from here on: the reference's own CommonAgent.__init__ body (common_agent.py:29-70),
    # executed statement by statement on the injected object.
    with _Quiet():
        A._load_config_params(config)
        A.is_discrete = False
        A._setup_action_space()
        A.bounds_loss_coef = config.get('bounds_loss_coef', None)
        A.clip_actions = config.get('clip_actions', True)
        A._save_intermediate = config.get('save_intermediate', False)
        net_config = A._build_net_config()
        A.network = model_cls(builder)
        A.model = A.network.build(net_config)
    A.model.to(device)
    A.states = None
    A.last_lr = float(A.last_lr)
    A.optimizer = torch.optim.Adam(A.model.parameters(), float(A.last_lr), eps=1e-08, weight_decay=A.weight_decay)
    if A.normalize_input:
        A.running_mean_std = RunningMeanStd((obs_size,)).to(device)
    A.dataset = ref.amp_datasets.AMPDataset(A.batch_size, A.minibatch_size, A.is_discrete, A.is_rnn, device, A.seq_len)

    if kind in ('amp', 'ase'):
        # AMPAgent.__init__ tail (amp_agent.py:25-26)
        if A._normalize_amp_input:
            A._amp_input_mean_std = RunningMeanStd(A._amp_observation_space.shape).to(device)

    # init_tensors(): experience buffer + reference-side extra buffers
    A.experience_buffer = _ExperienceBuffer(A.horizon_length, A.num_actors, (obs_size,), act_size, 1, device)
    A.update_list = ['actions', 'neglogpacs', 'values', 'mus', 'sigmas']
    A.tensor_list = A.update_list + ['obses', 'states', 'dones']
    A.vec_env = types.SimpleNamespace(env=types.SimpleNamespace(
        task=types.SimpleNamespace(num_envs=num_envs, viewer=None, progress_buf=torch.zeros(num_envs, dtype=torch.long)),
        fetch_amp_obs_demo=demo_fetch))
    td = A.experience_buffer.tensor_dict
    td['next_obses'] = torch.zeros_like(td['obses'])      # common_agent.py:76-79
    td['next_values'] = torch.zeros_like(td['values'])
    A.tensor_list += ['next_obses']
    if kind in ('amp', 'ase'):
        A._build_amp_buffers()                           # amp_agent.py:502-518 (reference code)
    if kind == 'ase':
        bs = A.experience_buffer.obs_base_shape           # ase_agent.py:20-27
        td['ase_latents'] = torch.zeros(bs + (A._latent_dim,), dtype=torch.float32, device=device)
        A._ase_latents = torch.zeros((bs[-1], A._latent_dim), dtype=torch.float32, device=device)
        A.tensor_list += ['ase_latents']
    return A


def record_sampled_latents(agent):
    """Wrap agent._sample_latents (instance attribute; reference code untouched) so every z drawn
    inside _diversity_loss (ase_agent.py:451) is recorded for injection into the HIP path."""
    rec = []
    orig = agent._sample_latents

    def wrapped(n):
        z = orig(n)
        rec.append(z.detach().clone())
        return z

    agent._sample_latents = wrapped
    return rec
