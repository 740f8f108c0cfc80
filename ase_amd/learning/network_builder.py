"""Network builders with the reference's builder contract and checkpoint layout.

Mirrors (paths under /root/reference/ase/):
  AMPBuilder  learning/amp_network_builder.py:11-125  (rl_games A2CBuilder MLP actor/critic + disc)
  ASEBuilder  learning/ase_network_builder.py:18-351  (style-conditioned actor, latent critic, encoder)
  HRLBuilder  learning/hrl_network_builder.py:8-39    (plain A2C MLP, tanh on mu)

``Builder.load(params)`` takes the yaml ``params.network`` subtree, ``Builder.build(name, **kw)``
takes ``actions_num, input_shape, num_seqs, value_size, amp_input_shape, ase_latent_shape`` and returns an
``nn.Module`` whose ``state_dict()`` has exactly the reference's keys / shapes / dtypes (including the
``_enc_mlp.*`` aliases of ``_disc_mlp.*`` when the encoder shares the discriminator trunk,
learning/ase_network_builder.py:202-203).  The module tree exists for naming and checkpoint I/O only:
every parameter is a view into ONE flat f32 buffer (``flat_params``), which is what the HIP engine,
the fused Adam kernel and the gradient all-reduce operate on.
"""
import math

import torch
import torch.nn as nn

DISC_LOGIT_INIT_SCALE = 1.0     # learning/amp_network_builder.py:9
ENC_LOGIT_INIT_SCALE = 0.1      # learning/ase_network_builder.py:12
STYLE_UNITS = [512, 256]        # learning/ase_network_builder.py:156
STYLE_INIT_RANGE = 1.0          # learning/ase_network_builder.py:327


def _mlp(in_size, units):
    layers = []
    for u in units:
        layers.append(nn.Linear(in_size, u))
        layers.append(nn.Identity())          # activation slot: keeps the child indices 0, 2, 4, ...
        in_size = u
    return nn.Sequential(*layers)


class _StyleCatNet(nn.Module):   # parameter names of AMPStyleCatNet1 (learning/ase_network_builder.py:273-351)
    def __init__(self, obs_size, latent, units):
        super().__init__()
        self._style_mlp = _mlp(latent, STYLE_UNITS)
        self._style_dense = nn.Linear(STYLE_UNITS[-1], latent)
        dense, k = [], obs_size + latent
        for u in units:
            dense.append(nn.Linear(k, u))
            k = u
        self._dense_layers = nn.ModuleList(dense)


class _LatentMLP(nn.Module):     # parameter names of AMPMLPNet (learning/ase_network_builder.py:232-271)
    def __init__(self, obs_size, latent, units):
        super().__init__()
        self._mlp = _mlp(obs_size + latent, units)


class A2CNetwork(nn.Module):
    """kind: 'amp' | 'ase' | 'ppo' (HRL high-level policy / plain A2C)."""

    def __init__(self, kind, params, actions_num, input_shape, value_size=1, num_seqs=1, amp_input_shape=None,
                 ase_latent_shape=None, device='cpu'):
        super().__init__()
        assert value_size == 1
        self.kind = kind
        self.actions_num = int(actions_num)
        self.obs_size = int(input_shape[-1])
        self.value_size = 1
        self.num_seqs = num_seqs
        self.is_continuous = True
        sp = params['space']['continuous']
        assert sp['fixed_sigma'] and not sp['learn_sigma'], "only the reference's frozen log-std is supported"
        assert sp.get('mu_activation', 'None') == 'None' and sp.get('sigma_activation', 'None') == 'None'
        assert params.get('separate', False), "reference configs use separate actor / critic trunks"
        self.units = list(params['mlp']['units'])
        self.activation = params['mlp']['activation']
        # rl_games activations_factory names with a HIP epilogue (csrc/act.h): value + first / second derivative
        SUPPORTED = ('relu', 'tanh', 'sigmoid', 'elu', 'selu', 'swish', 'gelu', 'softplus')
        for blk in ('mlp', 'disc', 'enc'):
            if blk in params:
                assert params[blk]['activation'] in SUPPORTED, f"{blk}.activation {params[blk]['activation']!r}: one of {SUPPORTED}"
        self.mu_tanh = kind == 'ppo'
        self.latent_dim = int(ase_latent_shape[-1]) if kind == 'ase' else 0
        self.style_units = list(STYLE_UNITS)
        self.amp_obs_size = int(amp_input_shape[-1]) if kind in ('amp', 'ase') else 0

        self.actor_cnn, self.critic_cnn = nn.Sequential(), nn.Sequential()
        if kind == 'ase':
            self.actor_mlp = _StyleCatNet(self.obs_size, self.latent_dim, self.units)
            self.critic_mlp = _LatentMLP(self.obs_size, self.latent_dim, self.units)
        else:
            self.actor_mlp = _mlp(self.obs_size, self.units)
            self.critic_mlp = _mlp(self.obs_size, self.units)
        out = self.units[-1]
        self.value = nn.Linear(out, 1)
        self.mu = nn.Linear(out, self.actions_num)
        self.sigma = nn.Parameter(torch.full((self.actions_num,), float(sp['sigma_init']['val'])), requires_grad=False)
        if kind in ('amp', 'ase'):
            self.disc_units = list(params['disc']['units'])
            self.disc_activation = params['disc']['activation']
            self._disc_mlp = _mlp(self.amp_obs_size, self.disc_units)
            self._disc_logits = nn.Linear(self.disc_units[-1], 1)
        self.enc_separate = False
        if kind == 'ase':
            self.enc_separate = bool(params['enc']['separate'])
            self.enc_units = list(params['enc']['units'])
            self.enc_activation = params['enc']['activation']
            if self.enc_separate:
                self._enc_mlp = _mlp(self.amp_obs_size, self.enc_units)
                enc_in = self.enc_units[-1]
            else:
                self._enc_mlp = self._disc_mlp          # same module under a second name (checkpoint aliases)
                enc_in = self.disc_units[-1]
            self._enc = nn.Linear(enc_in, self.latent_dim)

        # initialisers: rl_games 'default' = PyTorch's nn.Linear weight init; every bias zero
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.zeros_(m.bias)
        if kind == 'ase':
            nn.init.uniform_(self.actor_mlp._style_dense.weight, -STYLE_INIT_RANGE, STYLE_INIT_RANGE)
            nn.init.uniform_(self._enc.weight, -ENC_LOGIT_INIT_SCALE, ENC_LOGIT_INIT_SCALE)
        if kind in ('amp', 'ase'):
            nn.init.uniform_(self._disc_logits.weight, -DISC_LOGIT_INIT_SCALE, DISC_LOGIT_INIT_SCALE)
        self._flatten(torch.device(device))
        self.infer = None          # ase_amd.inference.InferenceEngine, attached lazily

    # ------------------------------------------------------------------ flat storage
    def _flatten(self, device):
        named = list(self.named_parameters())         # de-duplicated (shared trunk appears once)
        train = [(k, p) for k, p in named if p.requires_grad]
        frozen = [(k, p) for k, p in named if not p.requires_grad]
        total = sum(p.numel() for _, p in named)
        flat = torch.empty(total, dtype=torch.float32, device=device)
        self.param_slices = {}
        o = 0
        for k, p in train + frozen:
            n = p.numel()
            flat[o:o + n].copy_(p.detach().reshape(-1))
            p.data = flat[o:o + n].view(p.shape)
            self.param_slices[k] = (o, tuple(p.shape))
            o += n
        self.flat_params = flat
        self.trainable_numel = sum(p.numel() for _, p in train)

    def _apply(self, fn, recurse=True):
        # .to(device) / .float() would re-allocate every parameter separately: move the flat buffer instead
        flat = fn(self.flat_params)
        if flat.dtype != torch.float32:
            raise TypeError("master weights stay f32 (bf16 shadows are derived caches)")
        if flat is not self.flat_params:
            if getattr(self, '_engine_bound', False):
                raise RuntimeError("the parameters were moved after an UpdateEngine bound them (its shadows, gradient / "
                                   "Adam buffers and pointer tables refer to the old storage): move the network first")
            self.flat_params = flat
            for k, p in self.named_parameters():
                o, shp = self.param_slices[k]
                p.data = flat[o:o + math.prod(shp)].view(shp)
            self.infer = None
        return self

    # ------------------------------------------------------------------ reference network API
    def is_rnn(self):
        return False

    def is_separate_critic(self):
        return True

    def get_default_rnn_state(self):
        return None

    def _engine(self):
        if self.infer is None:
            from ..inference import InferenceEngine
            self.infer = InferenceEngine(self)
        return self.infer

    def forward(self, obs_dict):
        obs = obs_dict['obs']
        z = obs_dict.get('ase_latents') if self.kind == 'ase' else None
        mu, sigma = self.eval_actor(obs, z) if self.kind == 'ase' else self.eval_actor(obs)
        value = self.eval_critic(obs, z) if self.kind == 'ase' else self.eval_critic(obs)
        return mu, sigma, value, obs_dict.get('rnn_states', None)

    def eval_actor(self, obs, ase_latents=None, use_hidden_latents=False):
        assert not use_hidden_latents
        mu = self._engine().actor(obs, ase_latents)
        return mu, mu * 0.0 + self.sigma

    def eval_critic(self, obs, ase_latents=None, use_hidden_latents=False):
        return self._engine().critic(obs, ase_latents)

    def eval_disc(self, amp_obs):
        return self._engine().disc(amp_obs)

    def eval_enc(self, amp_obs):
        return self._engine().enc(amp_obs)

    def sample_latents(self, n):
        return self._engine().sample_latents(n)

    def get_disc_logit_weights(self):
        return torch.flatten(self._disc_logits.weight)

    def get_disc_weights(self):
        ws = [torch.flatten(m.weight) for m in self._disc_mlp.modules() if isinstance(m, nn.Linear)]
        ws.append(torch.flatten(self._disc_logits.weight))
        return ws

    def get_enc_weights(self):
        ws = [torch.flatten(m.weight) for m in self._enc_mlp.modules() if isinstance(m, nn.Linear)]
        ws.append(torch.flatten(self._enc.weight))
        return ws


class _Builder:
    kind = None

    def __init__(self, **kwargs):
        self.params = None

    def load(self, params):
        self.params = params

    def build(self, name, **kwargs):
        return A2CNetwork(self.kind, self.params, **kwargs)

    def __call__(self, name, **kwargs):
        return self.build(name, **kwargs)


class AMPBuilder(_Builder):
    kind = 'amp'


class ASEBuilder(_Builder):
    kind = 'ase'


class HRLBuilder(_Builder):
    kind = 'ppo'
