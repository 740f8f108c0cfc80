"""Generate the golden vectors under tests/golden/ by running the REFERENCE'S OWN code
(/root/reference/ase/learning/*.py through oracle/ref_runner.py).  TEST INFRASTRUCTURE ONLY.

    python oracle/make_golden.py            # (re)writes tests/golden/*.pt

Each case drives two full ``train_epoch`` calls of the reference agent on a seeded synthetic
rollout (ase_amd/synthetic.py) and records everything another implementation needs to replay
it bit-for-bit on the same inputs: initial weights, experience buffers, demo stream, every random
draw made inside the update (dataset permutations, replay/demo sample permutations, the latents
drawn by _diversity_loss) — and what the reference produced: per-step train_result, the
gradients of the first step, running statistics and final weights.

Only ``play_steps`` is replaced (it is the Isaac Gym loop): its replacement fills nothing and runs
the reference's own tail statements (learning/ase_agent.py:95-115) on the synthetic buffers.
"""
import copy
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from ase_amd.synthetic import EnvSpec, SyntheticSource  # noqa: E402
from oracle import ref_runner  # noqa: E402

OUT = os.environ.get("ASE_GOLDEN_OUT", os.path.join(os.path.dirname(HERE), "tests", "golden"))

NET_ASE = {
    'name': 'ase', 'separate': True,
    'space': {'continuous': {'mu_activation': 'None', 'sigma_activation': 'None', 'mu_init': {'name': 'default'},
                             'sigma_init': {'name': 'const_initializer', 'val': -2.9},
                             'fixed_sigma': True, 'learn_sigma': False}},
    'mlp': {'units': [48, 40, 24], 'activation': 'relu', 'd2rl': False, 'initializer': {'name': 'default'},
            'regularizer': {'name': 'None'}},
    'disc': {'units': [40, 24], 'activation': 'relu', 'initializer': {'name': 'default'}},
    'enc': {'units': [40, 24], 'activation': 'relu', 'separate': False, 'initializer': {'name': 'default'}},
}

CFG_ASE = {
    'name': 'Humanoid', 'env_name': 'rlgpu', 'multi_gpu': False, 'ppo': True, 'mixed_precision': False,
    'normalize_input': True, 'normalize_value': True, 'reward_shaper': {'scale_value': 1},
    'normalize_advantage': True, 'gamma': 0.99, 'tau': 0.95, 'learning_rate': 2e-5, 'lr_schedule': 'constant',
    'score_to_win': 20000, 'max_epochs': 100000, 'save_best_after': 50, 'save_frequency': 50,
    'print_stats': True, 'grad_norm': 1.0, 'entropy_coef': 0.0, 'truncate_grads': False, 'e_clip': 0.2,
    'horizon_length': 8, 'minibatch_size': 32, 'mini_epochs': 2, 'critic_coef': 5, 'clip_value': False,
    'seq_len': 4, 'bounds_loss_coef': 10, 'amp_obs_demo_buffer_size': 96, 'amp_replay_buffer_size': 160,
    'amp_replay_keep_prob': 0.01, 'amp_batch_size': 16, 'amp_minibatch_size': 16, 'disc_coef': 5,
    'disc_logit_reg': 0.01, 'disc_grad_penalty': 5, 'disc_reward_scale': 2, 'disc_weight_decay': 0.0001,
    'normalize_amp_input': True, 'enable_eps_greedy': True, 'latent_dim': 16, 'latent_steps_min': 1,
    'latent_steps_max': 6, 'amp_latent_grad_bonus': 0.0, 'amp_latent_grad_bonus_max': 100.0,
    'amp_diversity_bonus': 0.01, 'amp_diversity_tar': 1.0, 'enc_coef': 5, 'enc_weight_decay': 0.0,
    'enc_reward_scale': 1, 'enc_grad_penalty': 0, 'task_reward_w': 0.0, 'disc_reward_w': 0.5, 'enc_reward_w': 0.5,
}


def _yaml(name):
    import yaml
    with open(os.path.join(ref_runner.REFERENCE_ROOT, 'ase', 'data', 'cfg', 'train', 'rlg', name)) as f:
        return yaml.safe_load(f)['params']


def _case(kind):
    net = copy.deepcopy(NET_ASE)
    cfg = copy.deepcopy(CFG_ASE)
    if kind == 'ase_cfg2':
        # BASELINE.json configs[1] / SURVEY 8d 'Config 2' at the REAL layer widths: network and hyper-parameters of
        # ase/data/cfg/train/rlg/ase_humanoid.yaml:9-46,53-114 VERBATIM (read from the reference tree), with only the batch
        # geometry reduced so that the unmodified reference finishes in seconds on the host: 64 envs x horizon 32 = 2048
        # samples, minibatch 512 / amp minibatch 128, 2 mini-epochs = 8 optimisation steps; rings 1024 / 4096 rows (their
        # size only enters through the recorded sampling indices).
        y = _yaml('ase_humanoid.yaml')
        n, c = y['network'], y['config']
        c.update(minibatch_size=512, amp_minibatch_size=128, mini_epochs=2, amp_batch_size=128,
                 amp_obs_demo_buffer_size=1024, amp_replay_buffer_size=4096)
        return n, c
    if kind == 'ase_swish':
        # row X1: a non-ReLU activation (rl_games activations_factory 'swish' = SiLU, learning/ase_network_builder.py:162) in the
        # policy MLPs AND the discriminator / encoder trunk: the gradient penalty's double backward (learning/amp_agent.py:453-459)
        # then has a second-derivative term
        n, c = _case('ase')
        n['mlp']['activation'] = 'swish'
        n['disc']['activation'] = 'swish'
        n['enc']['activation'] = 'swish'
        return n, c
    if kind.startswith('ase_act_'):
        # the other curved activations of rl_games' factory (elu, gelu, softplus, selu, sigmoid) through the same double backward
        n, c = _case('ase')
        for part in ('mlp', 'disc', 'enc'):
            n[part]['activation'] = kind[len('ase_act_'):]
        return n, c
    if kind == 'hrl_cfg4':
        # BASELINE.json configs[3]: the HRL high-level policy's PPO update at its real shape 258 -> [1024, 512] -> 64 with
        # tanh(mu) (learning/hrl_network_builder.py:26-29), ase/data/cfg/train/rlg/hrl_humanoid.yaml:9-39,45-77 verbatim
        # but for the batch geometry (64 envs x 32, minibatch 512, 2 mini-epochs)
        y = _yaml('hrl_humanoid.yaml')
        n, c = y['network'], y['config']
        c.update(minibatch_size=512, mini_epochs=2)
        return n, c
    if kind == 'amp_cfg1':
        # BASELINE.json configs[0]: AMP agent, 64 envs x horizon 16, obs 253 / act 31, 2-layer [256, 128] MLPs, the other
        # hyper-parameters of ase/data/cfg/train/rlg/amp_humanoid.yaml (SURVEY §8d 'Config 1': minibatch 256, amp minibatch 64,
        # 6 mini-epochs = 24 steps).  The demo / replay rings are shrunk (512 / 2048 rows instead of 200000): their size only
        # enters through the sampling indices, which are recorded.
        n, c = _case('amp')
        n['mlp']['units'] = [256, 128]
        n['disc']['units'] = [256, 128]
        c.update(horizon_length=16, minibatch_size=256, mini_epochs=6, amp_minibatch_size=64, amp_batch_size=128,
                 amp_obs_demo_buffer_size=512, amp_replay_buffer_size=2048, amp_replay_keep_prob=0.01, disc_reward_scale=2,
                 task_reward_w=0.0, disc_reward_w=1.0, learning_rate=5e-5, disc_grad_penalty=5, disc_logit_reg=0.05,
                 disc_weight_decay=0.0001)
        return n, c
    if kind == 'amp':
        net['name'] = 'amp'
        del net['enc']
        net['mlp']['units'] = [48, 24]
        for k in ('latent_dim', 'latent_steps_min', 'latent_steps_max', 'amp_diversity_bonus', 'amp_diversity_tar',
                  'enc_coef', 'enc_weight_decay', 'enc_reward_scale', 'enc_grad_penalty', 'enc_reward_w'):
            cfg.pop(k)
        cfg['task_reward_w'], cfg['disc_reward_w'] = 0.5, 0.5
    elif kind == 'ppo':
        net['name'] = 'hrl'
        del net['enc'], net['disc']
        net['mlp']['units'] = [48, 24]
        net['space']['continuous']['sigma_init']['val'] = -2.3
    elif kind == 'ase_sep':
        net['enc']['separate'] = True
    elif kind in ('ase_gp', 'ase_sep_gp'):
        # the encoder's gradient penalty (learning/ase_agent.py:431-441) and weight decay switched on
        net['enc']['separate'] = kind == 'ase_sep_gp'
        cfg.update(enc_grad_penalty=5, enc_weight_decay=0.0001)
    return net, cfg


def _policy_from_agent(A, kind):
    """get_action_values / _eval_critic (learning/ase_agent.py:117-148,385-393) in eval mode."""
    def policy(obs, z):
        A.set_eval()
        with torch.no_grad():
            pobs = A._preproc_obs(obs)
            d = {'is_train': False, 'prev_actions': None, 'obs': pobs, 'rnn_states': None}
            if z is not None:
                d['ase_latents'] = z
            res = A.model(d)
            value = res['values']
            if A.normalize_value:
                value = A.value_mean_std(value, True)
            return res['mus'], res['sigmas'], value
    return policy


def _tail(A, kind):
    """The statements after the rollout loop, verbatim in effect (learning/ase_agent.py:95-115,
    learning/amp_agent.py:118-137, learning/common_agent.py:295-307), executed on A."""
    from rl_games.common import a2c_common
    td = A.experience_buffer.tensor_dict
    mb_fdones = td['dones'].float()
    mb_values = td['values']
    mb_next_values = td['next_values']
    mb_rewards = td['rewards']
    amp_rewards = {}
    if kind in ('amp', 'ase', 'ase_sep', 'ase_gp', 'ase_sep_gp'):
        if kind == 'amp':
            amp_rewards = A._calc_amp_rewards(td['amp_obs'])
        else:
            amp_rewards = A._calc_amp_rewards(td['amp_obs'], td['ase_latents'])
        mb_rewards = A._combine_rewards(mb_rewards, amp_rewards)
    mb_advs = A.discount_values(mb_fdones, mb_values, mb_rewards, mb_next_values)
    mb_returns = mb_advs + mb_values
    batch_dict = A.experience_buffer.get_transformed_list(a2c_common.swap_and_flatten01, A.tensor_list)
    batch_dict['returns'] = a2c_common.swap_and_flatten01(mb_returns)
    batch_dict['played_frames'] = A.batch_size
    for k, v in amp_rewards.items():
        batch_dict[k] = a2c_common.swap_and_flatten01(v)
    A._golden_tail = {'mb_advs': mb_advs.clone(), 'mb_returns': mb_returns.clone(),
                      **{k: v.clone() for k, v in amp_rewards.items()}}
    return batch_dict


def _rms_state(m):
    return {'mean': m.running_mean.clone(), 'var': m.running_var.clone(), 'count': m.count.clone()}


def make_case(name, kind, seed, num_envs=16, obs_size=37, act_size=7, amp_size=44, epochs=2, slim=False, regen=False,
              seeded=False, sample=0, warm=0):
    """slim: leave out what only the single-step tests read (first-step gradients / weights, the checkpoint dictionary);
    regen: leave out every tensor the test can regenerate from the seeded synthetic source (observations, AMP observations,
    demo stream - tests/test_agent_emu.py:regenerate) and the duplicated dataset rows."""
    akind = {'ase_sep': 'ase', 'amp_cfg1': 'amp', 'ase_gp': 'ase', 'ase_sep_gp': 'ase', 'ase_cfg2': 'ase', 'ase_swish': 'ase',
             **{'ase_act_' + a_: 'ase' for a_ in ('elu', 'gelu', 'softplus', 'selu', 'sigmoid')},
             'hrl_cfg4': 'ppo'}.get(kind, kind)
    net, cfg = _case(kind)
    akind_kind = kind
    spec = EnvSpec(num_envs=num_envs, horizon=cfg['horizon_length'], obs_size=obs_size, act_size=act_size,
                   amp_obs_size=amp_size if akind != 'ppo' else 0, latent_dim=cfg.get('latent_dim', 0),
                   latent_steps_min=cfg.get('latent_steps_min', 1), latent_steps_max=cfg.get('latent_steps_max', 2),
                   episode_length=20)
    src = SyntheticSource(spec, seed=1234 + seed)
    demo_log = []

    def demo_fetch(n):
        x = src.fetch_amp_obs_demo(n)
        demo_log.append(x.clone())
        return x

    torch.manual_seed(seed)
    A = ref_runner.build_ref_agent(akind, net, cfg, num_envs=num_envs, obs_size=obs_size, act_size=act_size,
                                   amp_obs_size=amp_size if akind != 'ppo' else None, demo_fetch=demo_fetch, seed=seed)
    init_sd = {k.replace('a2c_network.', '', 1): v.detach().clone() for k, v in A.model.state_dict().items()}
    if seeded:
        # seeded=True (real widths): the initial weights are a function of (shapes, the reference initialiser's own
        # bounds, seed) - tests/helpers.seeded_init - loaded into the reference agent here and regenerated by the tests
        from tests.helpers import seeded_init
        shapes = {k: tuple(v.shape) for k, v in init_sd.items()}
        bounds = {}
        for k, v in init_sd.items():
            if v.numel() == 0 or float(v.abs().max()) == float(v.abs().min()):
                bounds[k] = {'const': float(v.reshape(-1)[0]) if v.numel() else 0.0}
            else:
                bounds[k] = float('%.4g' % (float(v.abs().max()) * 1.0001))
        init_sd = seeded_init(shapes, bounds, seed)
        A.model.load_state_dict({'a2c_network.' + k: v for k, v in init_sd.items()})
    G = {'kind': akind, 'net': net, 'cfg': cfg, 'seed': seed,
         'spec': dict(num_envs=num_envs, obs_size=obs_size, act_size=act_size, amp_obs_size=amp_size),
         'init_sd': init_sd,
         'trainable': [k.replace('a2c_network.', '', 1) for k, p in A.model.named_parameters() if p.requires_grad],
         'epochs': []}
    if warm:
        # warm=n: the reference's own RunningMeanStd modules see n batches of 4096 observations (drawn from the synthetic
        # feature distributions with a SEPARATE generator) before the recorded epoch - the regime of real training, where one
        # more minibatch barely moves the statistics and the first step's importance ratio is ~1.  Cold statistics (count 1)
        # make the first minibatch re-normalise everything: clip fraction ~0.98, ratios of e^10 (kept as the stress seed).
        g2 = torch.Generator().manual_seed(seed * 31 + 4242)
        A.set_train()
        with torch.no_grad():
            for _ in range(warm):
                A.running_mean_std(src.obs_fs.draw(4096, g2))
                if akind != 'ppo':
                    A._amp_input_mean_std(src.amp_fs.draw(4096, g2))
        G['warm'] = warm
    if akind != 'ppo':
        A._init_amp_demo_buf()        # learning/amp_agent.py:520-528 (reference code, calls demo_fetch)
        G['demo_init'] = torch.cat(demo_log, 0)
        demo_log.clear()
        G['demo_sample_perm0'] = A._amp_obs_demo_buffer._sample_idx.clone()
        G['replay_sample_perm0'] = A._amp_replay_buffer._sample_idx.clone()
    G['dataset_perm0'] = A.dataset._idx_buf.clone()

    for ep in range(epochs):
        E = {}
        exp = src.experience(_policy_from_agent(A, akind), with_amp=akind != 'ppo', with_latents=akind == 'ase')
        td = A.experience_buffer.tensor_dict
        for k, v in exp.items():
            if k in td:
                td[k].copy_(v)
        E['exp'] = {k: v.clone() for k, v in exp.items()}
        E['rms_before'] = {'obs': _rms_state(A.running_mean_std), 'value': _rms_state(A.value_mean_std)}
        if akind != 'ppo':
            E['rms_before']['amp'] = _rms_state(A._amp_input_mean_std)

        # --- recorders (instance attributes only; the reference classes are untouched) ---
        perms = [A.dataset._idx_buf.clone()]
        orig_shuffle = A.dataset._shuffle_idx_buf

        def shuffle():
            orig_shuffle()
            perms.append(A.dataset._idx_buf.clone())
        A.dataset._shuffle_idx_buf = shuffle
        zs = ref_runner.record_sampled_latents(A) if akind == 'ase' else []
        steps = []
        first_grads = {}
        orig_calc = type(A).calc_gradients

        def calc(input_dict, ep=ep):
            if not steps and ep == 0:
                E['first_minibatch'] = {k: v.clone() for k, v in input_dict.items() if v is not None}
                E['rms_step0_before'] = {'obs': _rms_state(A.running_mean_std)}
                if akind != 'ppo':
                    E['rms_step0_before']['amp'] = _rms_state(A._amp_input_mean_std)
            orig_calc(A, input_dict)
            if not steps and ep == 0:
                for k, p in A.model.named_parameters():
                    if p.grad is not None:
                        first_grads[k.replace('a2c_network.', '', 1)] = p.grad.detach().clone()
                E['sd_after_step0'] = {k.replace('a2c_network.', '', 1): v.detach().clone()
                                       for k, v in A.model.state_dict().items()}
            steps.append({k: (v.detach().clone() if torch.is_tensor(v) else torch.tensor(float(v)))
                          for k, v in A.train_result.items()})
        A.calc_gradients = calc
        A.play_steps = lambda: _tail(A, akind if (kind in ('amp_cfg1', 'ase_cfg2', 'ase_swish', 'hrl_cfg4') or kind.startswith('ase_act_')) else kind)
        if akind != 'ppo':
            E['replay_total_before'] = A._amp_replay_buffer.get_total_count()
            E['replay_head_before'] = A._amp_replay_buffer._head
            E['demo_head_before'] = A._amp_obs_demo_buffer._head
            E['demo_sample_head'] = A._amp_obs_demo_buffer._sample_head
            E['replay_sample_head'] = A._amp_replay_buffer._sample_head
            E['demo_sample_perm'] = A._amp_obs_demo_buffer._sample_idx.clone()
            E['replay_sample_perm'] = A._amp_replay_buffer._sample_idx.clone()

        A.epoch_num += 1
        A.train_epoch()      # REFERENCE CODE: learning/amp_agent.py:181-264 / common_agent.py:172-242

        A.dataset._shuffle_idx_buf = orig_shuffle
        if akind == 'ase':
            del A._sample_latents
        del A.calc_gradients
        E['tail'] = A._golden_tail
        E['dataset'] = {k: v.clone() for k, v in A.dataset.values_dict.items() if v is not None}
        E['dataset_perms'] = perms
        E['new_zs'] = zs
        E['steps'] = steps
        E['first_grads'] = first_grads
        E['demo_fetched'] = torch.cat(demo_log, 0) if demo_log else None
        demo_log.clear()
        E['rms_after'] = {'obs': _rms_state(A.running_mean_std), 'value': _rms_state(A.value_mean_std)}
        if akind != 'ppo':
            E['rms_after']['amp'] = _rms_state(A._amp_input_mean_std)
            E['replay_data_after'] = A._amp_replay_buffer._data_buf['amp_obs'].clone()
            E['replay_head_after'] = A._amp_replay_buffer._head
        E['sd_after'] = {k.replace('a2c_network.', '', 1): v.detach().clone() for k, v in A.model.state_dict().items()}
        G['epochs'].append(E)

    # the reference's checkpoint dictionary after the two epochs (rl_games A2CBase.get_full_state_weights through
    # learning/amp_agent.py:47-52 get_stats_weights): what save() writes and restore() reads
    import copy
    if not slim:
        G['ckpt_after'] = copy.deepcopy(A.get_full_state_weights())
    if slim and not sample:
        for E in G['epochs']:
            for k in ('sd_after_step0', 'first_grads'):
                E.pop(k, None)
    if regen:
        G['regen'] = {'source_seed': 1234 + seed, 'episode_length': 20}
        G.pop('demo_init', None)
        for E in G['epochs']:
            for k in ('obses', 'next_obses', 'amp_obs', 'ase_latents', 'rewards', 'dones', 'rand_action_mask'):
                E['exp'].pop(k, None)
            for k in ('dataset', 'first_minibatch', 'demo_fetched', 'replay_data_after'):
                E.pop(k, None)

    if seeded:
        del G['init_sd']
        G.update(init_shapes=shapes, init_bounds=bounds, init_seed=seed)
    if sample:
        from tests.helpers import pack_sampled
        G['sample'] = {'n': sample, 'seed': seed}
        for E in G['epochs']:
            for key in ('first_grads', 'sd_after_step0', 'sd_after'):
                if key in E:
                    E[key] = {k: (pack_sampled(k, v, sample, seed) if v.numel() > sample else v) for k, v in E[key].items()}
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.pt')
    torch.save(G, path)
    print('wrote', path, '%.2f MB' % (os.path.getsize(path) / 1e6))


if __name__ == '__main__':
    torch.set_num_threads(4)
    if len(sys.argv) > 1 and sys.argv[1] == 'real':           # real layer widths from the reference's yaml files (compact fixtures)
        torch.set_num_threads(16)
        for s_ in (0, 1, 2):      # seeds 0, 1: warmed statistics (ratio ~ 1); seed 2: cold statistics (the stress regime)
            make_case('ase_cfg2_small' + ('' if s_ == 0 else f'_s{s_}'), 'ase_cfg2', seed=30 + s_, num_envs=64, obs_size=253,
                      act_size=31, amp_size=1400, epochs=1, regen=True, seeded=True, sample=4096, slim=True, warm=0 if s_ == 2 else 16)
        make_case('hrl_cfg4_small', 'hrl_cfg4', seed=40, num_envs=64, obs_size=258, act_size=64, amp_size=0, epochs=1,
                  regen=True, seeded=True, sample=4096, slim=True, warm=16)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'swish':
        make_case('ase_swish_tiny', 'ase_swish', seed=7, epochs=1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'acts':           # one compact fixture per further activation of the factory
        for i, act in enumerate(('elu', 'gelu', 'softplus', 'selu', 'sigmoid')):
            make_case(f'ase_{act}_tiny', 'ase_act_' + act, seed=50 + i, epochs=1, regen=True, seeded=True, slim=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'encgp':          # only the N4 cases (the others are unchanged)
        make_case('ase_gp_tiny', 'ase_gp', seed=5, epochs=1)
        make_case('ase_sep_gp_tiny', 'ase_sep_gp', seed=6, epochs=1)
        sys.exit(0)
    make_case('ase_tiny', 'ase', seed=0)
    make_case('amp_tiny', 'amp', seed=1)
    make_case('ppo_tiny', 'ppo', seed=2)
    make_case('ase_sep_tiny', 'ase_sep', seed=3)
    # more seeds of the ASE case (seeds {0, 1, 2}) and BASELINE config 1's exact shape, as slim fixtures
    make_case('ase_tiny_s1', 'ase', seed=11, slim=True)
    make_case('ase_tiny_s2', 'ase', seed=12, slim=True)
    make_case('ase_gp_tiny', 'ase_gp', seed=5, epochs=1)
    make_case('ase_sep_gp_tiny', 'ase_sep_gp', seed=6, epochs=1)
    make_case('amp_cfg1', 'amp_cfg1', seed=21, num_envs=64, obs_size=253, act_size=31, amp_size=1400, epochs=1, slim=True, regen=True)
