"""Pin oracle/restated.py (the travel-capable CPU restatement) against the golden vectors that
oracle/make_golden.py recorded from the reference's own code: two full train_epoch updates per
case, every per-step train_result, first-step gradients, running statistics, replay ring and
final weights.  fp32, same inputs, same injected random draws."""
import os

import pytest
import torch

from oracle import restated as R

CASES = ['ase_tiny', 'amp_tiny', 'ppo_tiny', 'ase_sep_tiny', 'ase_gp_tiny', 'ase_sep_gp_tiny']
SCALARS = ['entropy', 'b_loss', 'actor_loss', 'actor_clip_frac', 'critic_loss', 'kl', 'disc_loss',
           'disc_grad_penalty', 'disc_logit_loss', 'disc_agent_acc', 'disc_demo_acc', 'enc_loss', 'enc_grad_penalty',
           'amp_diversity_loss']


def _close(a, b, rtol=2e-5, atol=1e-6, what=''):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.allclose(a.double(), b.double(), rtol=rtol, atol=atol), \
        (what, float((a.double() - b.double()).abs().max()))


def _load_sd(G):
    sep = G['net'].get('enc', {}).get('separate', False)
    return R.canonical_sd(G['init_sd'], sep, requires_grad=G['trainable'])


def _rms_close(s, g, what):
    for k in ('mean', 'var', 'count'):
        _close(s[k], g[k], rtol=1e-6, atol=1e-9, what=f'{what}.{k}')


def replay_case(G, check=True):
    kind, cfg = G['kind'], G['cfg']
    sd = _load_sd(G)
    spec = G['spec']
    rms = {'obs': R.rms_new(spec['obs_size']), 'value': R.rms_new(1)}
    adam = R.adam_new()
    if kind != 'ppo':
        rms['amp'] = R.rms_new(spec['amp_obs_size'])
        demo = R.RingBuffer(cfg['amp_obs_demo_buffer_size'], spec['amp_obs_size'], G['demo_sample_perm0'])
        demo.store(G['demo_init'][:cfg['amp_obs_demo_buffer_size']]) if G['demo_init'].shape[0] <= cfg['amp_obs_demo_buffer_size'] \
            else [demo.store(c) for c in G['demo_init'].split(cfg['amp_batch_size'])]
        replay = R.RingBuffer(cfg['amp_replay_buffer_size'], spec['amp_obs_size'], G['replay_sample_perm0'])
    out = []
    for E in G['epochs']:
        exp = E['exp']
        for k in rms:
            if check:
                _rms_close(rms[k], E['rms_before'][k], f'rms_before.{k}')
        ds, tail = R.prepare_epoch(kind, sd, rms, exp, cfg)
        if check:
            _close(tail['mb_advs'], E['tail']['mb_advs'], what='gae advs')
            _close(tail['mb_returns'], E['tail']['mb_returns'], what='gae returns')
            for k in ('disc_rewards', 'enc_rewards'):
                if k in E['tail']:
                    _close(tail[k], E['tail'][k], what=k)
            for k in ('advantages', 'old_values', 'returns', 'old_logp_actions', 'obs', 'actions', 'mu', 'sigma'):
                _close(ds[k], E['dataset'][k], what=f'dataset.{k}')
        if kind != 'ppo':
            B = ds['obs'].shape[0]
            demo.store(E['demo_fetched'])                                   # _update_amp_demos
            demo.sample_idx, demo.sample_head = E['demo_sample_perm'].clone(), E['demo_sample_head']
            assert demo.head == (E['demo_head_before'] + E['demo_fetched'].shape[0]) % demo.size
            ds['amp_obs_demo'] = demo.data[demo.sample_indices(B, next_perm=E['demo_sample_perm'])]
            if replay.total == 0:                                           # amp_agent.py:199-202
                ds['amp_obs_replay'] = ds['amp_obs']
            else:
                replay.sample_idx, replay.sample_head = E['replay_sample_perm'].clone(), E['replay_sample_head']
                ds['amp_obs_replay'] = replay.data[replay.sample_indices(B, next_perm=E['replay_sample_perm'])]
            if check:
                _close(ds['amp_obs_demo'], E['dataset']['amp_obs_demo'], what='demo sample')
                _close(ds['amp_obs_replay'], E['dataset']['amp_obs_replay'], what='replay sample')
        results = R.run_update(kind, sd, rms, adam, ds, cfg, E['dataset_perms'], E['new_zs'] or None)
        if kind != 'ppo':
            replay.store(ds['amp_obs'])                                     # _store_replay_amp_obs
        out.append(results)
        if not check:
            continue
        assert len(results) == len(E['steps'])
        for i, (r, g) in enumerate(zip(results, E['steps'])):
            for k in SCALARS:
                if k in g:
                    _close(r[k], g[k], rtol=1e-4, atol=1e-6, what=f'step{i}.{k}')
            for k in ('disc_agent_logit', 'disc_demo_logit'):
                if k in g:
                    _close(r[k], g[k], rtol=1e-4, atol=1e-5, what=f'step{i}.{k}')
        for k, g in E['sd_after'].items():
            if k not in sd:
                continue
            # Adam's m/(sqrt(v)+eps) is sign-like for tiny |g|: compare with an lr-scaled atol
            _close(sd[k].detach(), g, rtol=1e-5, atol=cfg['learning_rate'] * 0.02, what=f'sd_after.{k}')
        for k in rms:
            _rms_close(rms[k], E['rms_after'][k], f'rms_after.{k}')
        if kind != 'ppo':
            _close(replay.data, E['replay_data_after'], what='replay ring')
            assert replay.head == E['replay_head_after']
    return sd, rms, out


@pytest.mark.parametrize('name', CASES)
def test_restated_matches_reference_golden(name, golden_dir):
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    replay_case(G)


@pytest.mark.parametrize('name', CASES)
def test_first_step_gradients(name, golden_dir):
    """Every gradient tensor of the first optimisation step vs the reference's autograd."""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    kind, cfg, E = G['kind'], G['cfg'], G['epochs'][0]
    sd = _load_sd(G)
    rms = {k: {'mean': v['mean'].clone(), 'var': v['var'].clone(), 'count': v['count'].clone()}
           for k, v in E['rms_step0_before'].items()}
    mb = dict(E['first_minibatch'])
    z = E['new_zs'][0] if E['new_zs'] else None
    R.calc_gradients(kind, sd, rms, mb, cfg, z)
    assert set(E['first_grads']) == set(G['trainable'])
    for k, g in E['first_grads'].items():
        scale = float(g.abs().max()) + 1e-12
        _close(sd[k].grad, g, rtol=1e-4, atol=1e-5 * scale, what=f'grad.{k}')


def test_amp_obs_restatement_matches_reference(golden_dir):
    """N2 oracle (oracle/amp_obs.py) against the reference's own build_amp_observations (golden: oracle/make_golden_amp_obs.py),
    all four (local_root_obs, root_height_obs) settings, incl. the zero / tiny / wrapped exponential-map branches."""
    from oracle import amp_obs as A
    G = torch.load(os.path.join(golden_dir, 'amp_obs.pt'), weights_only=False)
    i = G['inputs']
    for (local_root, root_h), ref in G['outputs'].items():
        out = A.build_amp_observations(i['root_pos'], i['root_rot'], i['root_vel'], i['root_ang_vel'], i['dof_pos'], i['dof_vel'],
                                       i['key_body_pos'], local_root, root_h, G['dof_offsets'])
        assert out.shape == ref.shape == (96, 140)
        assert float((out - ref).abs().max()) <= 2e-6
    h = torch.arange(2 * 3 * 4, dtype=torch.float32).view(2, 3, 4)
    h2 = A.push_history(h.clone(), torch.full((2, 4), -1.0))
    assert torch.equal(h2[:, 1:], h[:, :2]) and torch.equal(h2[:, 0], torch.full((2, 4), -1.0))


def test_motion_state_restatement_matches_reference(golden_dir):
    """N2 oracle of the motion-clip sampler against the reference's own MotionLib.get_motion_state on two shipped clips
    (golden: oracle/make_golden_motion.py), incl. clip start / exact end / past the end."""
    from oracle import amp_obs as A
    G = torch.load(os.path.join(golden_dir, 'motion_state.pt'), weights_only=False)
    out = A.motion_state(G['clips'], G['motion_ids'], G['times'])
    for k, o in zip(G['outputs'], out):
        assert o.shape == G['outputs'][k].shape
        assert float((o - G['outputs'][k]).abs().max()) <= 1e-6, k


def test_rl_games_restatements_against_closed_forms():
    """The rl_games 1.1.4 pieces the reference calls (absent from /root/reference: parity unpinned at that boundary) anchored to
    what CAN be checked here - torch.distributions' closed forms and the algebra of the running-statistics merge:
      neglogp            = -Normal(mu, sigma).log_prob(x).sum(-1)                                  (exact)
      policy_kl          = KL(N(mu0, s0) || N(mu1, s1)).sum(-1).mean() up to its two 1e-5 guards   (sigma ~ 1: 1e-4)
      RunningMeanStd     mean after k merges = (the pseudo-sample 0 of the initial count 1 + every row) / (1 + rows); the
                         normaliser clamps to +-5 and un-normalise inverts normalise inside the clamp."""
    g = torch.Generator().manual_seed(7)
    mu, mu1 = torch.randn(64, 31, generator=g), torch.randn(64, 31, generator=g)
    ls, ls1 = torch.randn(31, generator=g) * 0.2, torch.randn(31, generator=g) * 0.2
    s, s1 = torch.exp(ls).expand(64, 31), torch.exp(ls1).expand(64, 31)
    x = mu + s * torch.randn(64, 31, generator=g)
    N = torch.distributions.Normal
    assert torch.allclose(R.neglogp(x, mu, s, ls.expand(64, 31)), -N(mu, s).log_prob(x).sum(-1), rtol=1e-6, atol=1e-5)
    kl_ref = torch.distributions.kl_divergence(N(mu, s), N(mu1, s1)).sum(-1).mean()
    assert abs(float(R.policy_kl(mu, s, mu1, s1)) - float(kl_ref)) <= 1e-4 * abs(float(kl_ref))
    assert abs(float(R.policy_kl(mu, s, mu, s))) <= 31 * 2e-5                       # (KL of a policy with itself: the guards' size)
    st = R.rms_new(5)
    rows = [torch.randn(n, 5, generator=g, dtype=torch.float64) * 3 + 1 for n in (7, 1000, 33)]
    for b in rows:
        R.rms_update(st, b)
    allx = torch.cat(rows)
    assert float(st['count']) == 1 + allx.shape[0]
    assert torch.allclose(st['mean'], allx.sum(0) / (1 + allx.shape[0]), rtol=1e-12, atol=1e-12)
    # variance: the merge treats every batch's UNBIASED variance as if it were its population variance (rl_games): between the two
    pop = ((torch.cat([torch.zeros(1, 5, dtype=torch.float64), allx]) - st['mean']) ** 2).mean(0)
    assert torch.all((st['var'] - pop).abs() <= 0.02 * pop + 1.0 / allx.shape[0])
    y = torch.randn(200, 5, generator=g, dtype=torch.float64) * 3 + 1
    z = R.rms_normalize(st, y)
    assert float(z.abs().max()) <= 5.0
    inside = z.abs() < 5.0
    assert torch.allclose(R.rms_unnormalize(st, z)[inside], y[inside], rtol=1e-9, atol=1e-9)
