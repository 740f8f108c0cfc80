"""rl_games.algos_torch.network_builder (1.1.4) — NetworkBuilder / A2CBuilder, restated for the
MLP-only, separate-critic, fixed-sigma continuous case every reference config uses
(amp_network_builder.py:16-18, ase_network_builder.py:32-34, hrl_network_builder.py:13-15)."""
import torch
import torch.nn as nn


class _Factory:
    def __init__(self):
        self._builders = {}

    def register_builder(self, name, builder):
        self._builders[name] = builder

    def create(self, name, **kwargs):
        return self._builders[name](**kwargs)


def _create_initializer(func, **kwargs):
    return lambda v: func(v, **kwargs)


class NetworkBuilder:
    def __init__(self, **kwargs):
        pass

    def load(self, params):
        pass

    class BaseNetwork(nn.Module):
        def __init__(self, **kwargs):
            nn.Module.__init__(self, **kwargs)
            af = _Factory()
            af.register_builder('relu', lambda **kw: nn.ReLU(**kw))
            af.register_builder('tanh', lambda **kw: nn.Tanh(**kw))
            af.register_builder('sigmoid', lambda **kw: nn.Sigmoid(**kw))
            af.register_builder('elu', lambda **kw: nn.ELU(**kw))
            af.register_builder('selu', lambda **kw: nn.SELU(**kw))
            af.register_builder('swish', lambda **kw: nn.SiLU(**kw))
            af.register_builder('gelu', lambda **kw: nn.GELU(**kw))
            af.register_builder('softplus', lambda **kw: nn.Softplus(**kw))
            af.register_builder('None', lambda **kw: nn.Identity())
            self.activations_factory = af

            inf = _Factory()
            inf.register_builder('const_initializer', lambda **kw: _create_initializer(nn.init.constant_, **kw))
            inf.register_builder('orthogonal_initializer', lambda **kw: _create_initializer(nn.init.orthogonal_, **kw))
            inf.register_builder('glorot_normal_initializer', lambda **kw: _create_initializer(nn.init.xavier_normal_, **kw))
            inf.register_builder('glorot_uniform_initializer', lambda **kw: _create_initializer(nn.init.xavier_uniform_, **kw))
            inf.register_builder('random_uniform_initializer', lambda **kw: _create_initializer(nn.init.uniform_, **kw))
            inf.register_builder('kaiming_normal', lambda **kw: _create_initializer(nn.init.kaiming_normal_, **kw))
            inf.register_builder('orthogonal', lambda **kw: _create_initializer(nn.init.orthogonal_, **kw))
            inf.register_builder('default', lambda **kw: nn.Identity())
            self.init_factory = inf

        def is_separate_critic(self):
            return False

        def is_rnn(self):
            return False

        def get_default_rnn_state(self):
            return None

        def _build_mlp(self, input_size, units, activation, dense_func,
                       norm_only_first_layer=False, norm_func_name=None, d2rl=False):
            assert not d2rl and norm_func_name is None
            in_size = input_size
            layers = []
            for unit in units:
                layers.append(dense_func(in_size, unit))
                layers.append(self.activations_factory.create(activation))
                in_size = unit
            return nn.Sequential(*layers)


class A2CBuilder(NetworkBuilder):
    def __init__(self, **kwargs):
        NetworkBuilder.__init__(self)

    def load(self, params):
        self.params = params

    class Network(NetworkBuilder.BaseNetwork):
        def __init__(self, params, **kwargs):
            actions_num = kwargs.pop('actions_num')
            input_shape = kwargs.pop('input_shape')
            self.value_size = kwargs.pop('value_size', 1)
            self.num_seqs = kwargs.pop('num_seqs', 1)
            NetworkBuilder.BaseNetwork.__init__(self)
            self.load(params)
            self.actor_cnn = nn.Sequential()
            self.critic_cnn = nn.Sequential()
            self.actor_mlp = nn.Sequential()
            self.critic_mlp = nn.Sequential()

            in_mlp_shape = input_shape[0]
            out_size = self.units[-1] if len(self.units) > 0 else in_mlp_shape
            mlp_args = {
                'input_size': in_mlp_shape,
                'units': self.units,
                'activation': self.activation,
                'norm_func_name': self.normalization,
                'dense_func': torch.nn.Linear,
                'd2rl': self.is_d2rl,
                'norm_only_first_layer': self.norm_only_first_layer,
            }
            self.actor_mlp = self._build_mlp(**mlp_args)
            if self.separate:
                self.critic_mlp = self._build_mlp(**mlp_args)

            self.value = torch.nn.Linear(out_size, self.value_size)
            self.value_act = self.activations_factory.create(self.value_activation)

            if self.is_continuous:
                self.mu = torch.nn.Linear(out_size, actions_num)
                self.mu_act = self.activations_factory.create(self.space_config['mu_activation'])
                mu_init = self.init_factory.create(**self.space_config['mu_init'])
                self.sigma_act = self.activations_factory.create(self.space_config['sigma_activation'])
                sigma_init = self.init_factory.create(**self.space_config['sigma_init'])
                if self.space_config['fixed_sigma']:
                    self.sigma = nn.Parameter(torch.zeros(actions_num, requires_grad=True, dtype=torch.float32),
                                              requires_grad=True)
                else:
                    self.sigma = torch.nn.Linear(out_size, actions_num)

            mlp_init = self.init_factory.create(**self.initializer)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    mlp_init(m.weight)
                    if getattr(m, "bias", None) is not None:
                        torch.nn.init.zeros_(m.bias)

            if self.is_continuous:
                mu_init(self.mu.weight)
                if self.space_config['fixed_sigma']:
                    sigma_init(self.sigma)
                else:
                    sigma_init(self.sigma.weight)

        def forward(self, obs_dict):
            obs = obs_dict['obs']
            states = obs_dict.get('rnn_states', None)
            a_out = self.actor_cnn(obs)
            a_out = a_out.contiguous().view(a_out.size(0), -1)
            c_out = self.critic_cnn(obs)
            c_out = c_out.contiguous().view(c_out.size(0), -1)
            a_out = self.actor_mlp(a_out)
            c_out = self.critic_mlp(c_out)
            value = self.value_act(self.value(c_out))
            mu = self.mu_act(self.mu(a_out))
            if self.space_config['fixed_sigma']:
                sigma = mu * 0.0 + self.sigma_act(self.sigma)
            else:
                sigma = self.sigma_act(self.sigma(a_out))
            return mu, sigma, value, states

        def is_separate_critic(self):
            return self.separate

        def load(self, params):
            self.separate = params.get('separate', False)
            self.units = params['mlp']['units']
            self.activation = params['mlp']['activation']
            self.initializer = params['mlp']['initializer']
            self.is_d2rl = params['mlp'].get('d2rl', False)
            self.norm_only_first_layer = params['mlp'].get('norm_only_first_layer', False)
            self.value_activation = params.get('value_activation', 'None')
            self.normalization = params.get('normalization', None)
            self.has_rnn = 'rnn' in params
            self.has_space = 'space' in params
            self.central_value = params.get('central_value', False)
            self.joint_obs_actions_config = params.get('joint_obs_actions', None)
            if self.has_space:
                self.is_multi_discrete = 'multi_discrete' in params['space']
                self.is_discrete = 'discrete' in params['space']
                self.is_continuous = 'continuous' in params['space']
                if self.is_continuous:
                    self.space_config = params['space']['continuous']
            else:
                self.is_discrete = False
                self.is_continuous = False
                self.is_multi_discrete = False
            self.has_cnn = 'cnn' in params

    def build(self, name, **kwargs):
        return A2CBuilder.Network(self.params, **kwargs)
