"""Hyper-parameters of the reference agents (the shape / coefficient contract of SURVEY.md §2 #14), as Python data.

Values follow the reference's ``ase/data/cfg/train/rlg/{ase,amp,hrl}_humanoid.yaml`` (cited per block); the
dictionaries have the layout rl_games hands to builders (``network``) and agents (``config``), so a yaml file of
the reference can be passed instead (``yaml.safe_load(...)["params"]``).  Note PyYAML reads ``2e-5`` as a string:
``normalize()`` fixes the numeric fields."""

import copy


# ---- ASE: ase/data/cfg/train/rlg/ase_humanoid.yaml
ASE_NETWORK = {'name': 'ase',
 'separate': True,
 'space': {'continuous': {'mu_activation': 'None',
                          'sigma_activation': 'None',
                          'mu_init': {'name': 'default'},
                          'sigma_init': {'name': 'const_initializer', 'val': -2.9},
                          'fixed_sigma': True,
                          'learn_sigma': False}},
 'mlp': {'units': [1024, 1024, 512],
         'activation': 'relu',
         'd2rl': False,
         'initializer': {'name': 'default'},
         'regularizer': {'name': 'None'}},
 'disc': {'units': [1024, 1024, 512], 'activation': 'relu', 'initializer': {'name': 'default'}},
 'enc': {'units': [1024, 512], 'activation': 'relu', 'separate': False, 'initializer': {'name': 'default'}}}

ASE_CONFIG = {'name': 'Humanoid',
 'env_name': 'rlgpu',
 'multi_gpu': False,
 'ppo': True,
 'mixed_precision': False,
 'normalize_input': True,
 'normalize_value': True,
 'reward_shaper': {'scale_value': 1},
 'normalize_advantage': True,
 'gamma': 0.99,
 'tau': 0.95,
 'learning_rate': 2e-05,
 'lr_schedule': 'constant',
 'score_to_win': 20000,
 'max_epochs': 100000,
 'save_best_after': 50,
 'save_frequency': 50,
 'print_stats': True,
 'grad_norm': 1.0,
 'entropy_coef': 0.0,
 'truncate_grads': False,
 'e_clip': 0.2,
 'horizon_length': 32,
 'minibatch_size': 16384,
 'mini_epochs': 6,
 'critic_coef': 5,
 'clip_value': False,
 'seq_len': 4,
 'bounds_loss_coef': 10,
 'amp_obs_demo_buffer_size': 200000,
 'amp_replay_buffer_size': 200000,
 'amp_replay_keep_prob': 0.01,
 'amp_batch_size': 512,
 'amp_minibatch_size': 4096,
 'disc_coef': 5,
 'disc_logit_reg': 0.01,
 'disc_grad_penalty': 5,
 'disc_reward_scale': 2,
 'disc_weight_decay': 0.0001,
 'normalize_amp_input': True,
 'enable_eps_greedy': True,
 'latent_dim': 64,
 'latent_steps_min': 1,
 'latent_steps_max': 150,
 'amp_latent_grad_bonus': 0.0,
 'amp_latent_grad_bonus_max': 100.0,
 'amp_diversity_bonus': 0.01,
 'amp_diversity_tar': 1.0,
 'enc_coef': 5,
 'enc_weight_decay': 0.0,
 'enc_reward_scale': 1,
 'enc_grad_penalty': 0,
 'task_reward_w': 0.0,
 'disc_reward_w': 0.5,
 'enc_reward_w': 0.5}

# ---- AMP: ase/data/cfg/train/rlg/amp_humanoid.yaml
AMP_NETWORK = {'name': 'amp',
 'separate': True,
 'space': {'continuous': {'mu_activation': 'None',
                          'sigma_activation': 'None',
                          'mu_init': {'name': 'default'},
                          'sigma_init': {'name': 'const_initializer', 'val': -2.9},
                          'fixed_sigma': True,
                          'learn_sigma': False}},
 'mlp': {'units': [1024, 512],
         'activation': 'relu',
         'd2rl': False,
         'initializer': {'name': 'default'},
         'regularizer': {'name': 'None'}},
 'disc': {'units': [1024, 512], 'activation': 'relu', 'initializer': {'name': 'default'}}}

AMP_CONFIG = {'name': 'Humanoid',
 'env_name': 'rlgpu',
 'multi_gpu': False,
 'ppo': True,
 'mixed_precision': False,
 'normalize_input': True,
 'normalize_value': True,
 'reward_shaper': {'scale_value': 1},
 'normalize_advantage': True,
 'gamma': 0.99,
 'tau': 0.95,
 'learning_rate': 2e-05,
 'lr_schedule': 'constant',
 'score_to_win': 20000,
 'max_epochs': 10000,
 'save_best_after': 50,
 'save_frequency': 50,
 'print_stats': True,
 'grad_norm': 1.0,
 'entropy_coef': 0.0,
 'truncate_grads': False,
 'e_clip': 0.2,
 'horizon_length': 32,
 'minibatch_size': 16384,
 'mini_epochs': 6,
 'critic_coef': 5,
 'clip_value': False,
 'seq_len': 4,
 'bounds_loss_coef': 10,
 'amp_obs_demo_buffer_size': 200000,
 'amp_replay_buffer_size': 200000,
 'amp_replay_keep_prob': 0.01,
 'amp_batch_size': 512,
 'amp_minibatch_size': 4096,
 'disc_coef': 5,
 'disc_logit_reg': 0.01,
 'disc_grad_penalty': 5,
 'disc_reward_scale': 2,
 'disc_weight_decay': 0.0001,
 'normalize_amp_input': True,
 'enable_eps_greedy': False,
 'task_reward_w': 0.0,
 'disc_reward_w': 1.0}

# ---- HRL: ase/data/cfg/train/rlg/hrl_humanoid.yaml
HRL_NETWORK = {'name': 'hrl',
 'separate': True,
 'space': {'continuous': {'mu_activation': 'None',
                          'sigma_activation': 'None',
                          'mu_init': {'name': 'default'},
                          'sigma_init': {'name': 'const_initializer', 'val': -2.3},
                          'fixed_sigma': True,
                          'learn_sigma': False}},
 'mlp': {'units': [1024, 512],
         'activation': 'relu',
         'd2rl': False,
         'initializer': {'name': 'default'},
         'regularizer': {'name': 'None'}}}

HRL_CONFIG = {'name': 'Humanoid',
 'env_name': 'rlgpu',
 'multi_gpu': False,
 'ppo': True,
 'mixed_precision': False,
 'normalize_input': True,
 'normalize_value': True,
 'reward_shaper': {'scale_value': 1},
 'normalize_advantage': True,
 'gamma': 0.99,
 'tau': 0.95,
 'learning_rate': 2e-05,
 'lr_schedule': 'constant',
 'score_to_win': 20000,
 'max_epochs': 10000,
 'save_best_after': 10,
 'save_frequency': 50,
 'print_stats': True,
 'grad_norm': 1.0,
 'entropy_coef': 0.0,
 'truncate_grads': False,
 'e_clip': 0.2,
 'horizon_length': 32,
 'minibatch_size': 16384,
 'mini_epochs': 6,
 'critic_coef': 5,
 'clip_value': False,
 'seq_len': 4,
 'bounds_loss_coef': 10,
 'task_reward_w': 0.9,
 'disc_reward_w': 0.1,
 'llc_steps': 5,
 'llc_config': 'ase/data/cfg/train/rlg/ase_humanoid_hrl.yaml'}


def normalize(cfg):
    """Numeric fields that yaml may deliver as strings."""
    cfg = copy.deepcopy(cfg)
    for k in ('learning_rate', 'disc_weight_decay', 'enc_weight_decay', 'disc_logit_reg'):
        if k in cfg:
            cfg[k] = float(cfg[k])
    return cfg


def get(kind):
    """(network params, agent config) deep copies for kind in {'ase', 'amp', 'hrl'}."""
    net, cfg = {'ase': (ASE_NETWORK, ASE_CONFIG), 'amp': (AMP_NETWORK, AMP_CONFIG), 'hrl': (HRL_NETWORK, HRL_CONFIG)}[kind]
    return copy.deepcopy(net), normalize(cfg)


# ---- one precision resolver for trainers and players (the same checkpoint config must give the same numerics in both)
_PRECISION_DTYPES = {'bf16': 'bfloat16', 'f16': 'float16', 'f16gp32': 'float16', 'f16gpx3': 'float16', 'f32': 'float32',
                     'bf16x3': 'float32'}


def resolve_precision(config):
    """config -> (precision name, torch dtype of the storage / MFMA type).  `precision` wins; without it the reference's
    `mixed_precision: True` (torch.cuda.amp autocast + GradScaler, learning/ase_agent.py:216,271-288) selects 'f16', else
    'bf16'.  Unknown names raise - a player must never fall back to another arithmetic silently."""
    import torch
    precision = config.get('precision', 'f16' if config.get('mixed_precision', False) else 'bf16')
    if precision not in _PRECISION_DTYPES:
        raise ValueError(f"unknown precision {precision!r}; one of {sorted(_PRECISION_DTYPES)}")
    return precision, getattr(torch, _PRECISION_DTYPES[precision])
