"""Model wrappers with the rl_games model contract the reference extends
(learning/amp_models.py:4-36, learning/ase_models.py:3-27, learning/hrl_models.py:4-17 under
/root/reference/ase): ``Model(network_builder).build(config) -> Network(a2c_network)`` whose
``forward(input_dict)`` returns the same dict keys.  All tensors come from the HIP inference path
(no autograd: the training step is ``UpdateEngine.step``)."""
import math

import torch
import torch.nn as nn


class _Network(nn.Module):
    with_disc = False
    with_enc = False

    def __init__(self, a2c_network):
        super().__init__()
        self.a2c_network = a2c_network

    def is_rnn(self):
        return False

    def get_default_rnn_state(self):
        return None

    @staticmethod
    def neglogp(x, mean, std, logstd):
        return 0.5 * (((x - mean) / std) ** 2).sum(dim=-1) + 0.5 * math.log(2.0 * math.pi) * x.size()[-1] \
            + logstd.sum(dim=-1)

    @torch.no_grad()
    def forward(self, input_dict):
        is_train = input_dict.get('is_train', True)
        prev_actions = input_dict.get('prev_actions', None)
        mu, logstd, value, states = self.a2c_network(input_dict)
        sigma = torch.exp(logstd)
        if is_train:
            entropy = (0.5 + 0.5 * math.log(2 * math.pi) + logstd).sum(dim=-1)
            result = {'prev_neglogp': torch.squeeze(self.neglogp(prev_actions, mu, sigma, logstd)), 'values': value,
                      'entropy': entropy, 'rnn_states': states, 'mus': mu, 'sigmas': sigma}
            if self.with_disc:
                net = self.a2c_network
                result['disc_agent_logit'] = net.eval_disc(input_dict['amp_obs'])
                result['disc_agent_replay_logit'] = net.eval_disc(input_dict['amp_obs_replay'])
                result['disc_demo_logit'] = net.eval_disc(input_dict['amp_obs_demo'])
            if self.with_enc:
                result['enc_pred'] = self.a2c_network.eval_enc(input_dict['amp_obs'])
            return result
        action = mu + sigma * torch.randn_like(mu)
        return {'neglogpacs': torch.squeeze(self.neglogp(action, mu, sigma, logstd)), 'values': value, 'actions': action,
                'rnn_states': states, 'mus': mu, 'sigmas': sigma}


class _Model:
    net_name = 'a2c'
    network_cls = _Network

    def __init__(self, network):
        self.network_builder = network

    def build(self, config):
        net = self.network_builder.build(self.net_name, **config)
        return self.network_cls(net)


class ModelAMPContinuous(_Model):
    net_name = 'amp'

    class Network(_Network):
        with_disc = True
    network_cls = Network


class ModelASEContinuous(_Model):
    net_name = 'ase'

    class Network(_Network):
        with_disc = True
        with_enc = True
    network_cls = Network


class ModelHRLContinuous(_Model):
    net_name = 'amp'       # the reference builds the HLC net under this name too (learning/hrl_models.py:9)

    class Network(_Network):
        pass
    network_cls = Network
