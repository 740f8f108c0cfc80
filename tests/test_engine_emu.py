"""Host logic of the update engine (buffer layout, launch sequence, analytic backward incl. the
gradient-penalty chain, concat-column maps, Adam, running statistics) checked on CPU: the engine
drives tests/emu_backend.py (same op semantics as the HIP kernels) and must reproduce what the
REFERENCE produced for the same minibatch (golden vectors): every loss scalar, every gradient
tensor of the first step and the post-Adam weights."""
import copy
import os

import pytest
import torch

from ase_amd.engine import UpdateEngine
from tests.emu_backend import EmuBackend
from tests.helpers import build_net, close, get_rms, set_rms

CASES = ['ase_tiny', 'amp_tiny', 'ppo_tiny', 'ase_sep_tiny', 'ase_gp_tiny', 'ase_sep_gp_tiny',
         'ase_swish_tiny']       # swish (SiLU) in the policy MLPs, the discriminator and the encoder: the curved gradient penalty


def first_step(G, be, dtype, device='cpu', grad_scale=None, engine_opts=None, gp_f32=False):
    kind, cfg, E = G['kind'], dict(G['cfg']), G['epochs'][0]
    if engine_opts:
        cfg['engine_opts'] = engine_opts
    if gp_f32:
        cfg['gp_f32'] = True
    net = build_net(G, device)
    mb = {k: v.to(device) for k, v in E['first_minibatch'].items()}
    M = mb['obs'].shape[0]
    amb = cfg.get('amp_minibatch_size', 0) if kind != 'ppo' else 0
    eng = UpdateEngine(kind, net, cfg, be, minibatch=M, amp_minibatch=amb, dtype=dtype, grad_scale=grad_scale)
    set_rms(eng.obs_state, E['rms_step0_before']['obs'])
    if kind != 'ppo':
        set_rms(eng.amp_state, E['rms_step0_before']['amp'])
    idx = torch.arange(M, dtype=torch.int32, device=device)
    streams = None
    if kind != 'ppo':
        streams = [(mb['amp_obs'], idx, (0, 0)), (mb['amp_obs_replay'], idx, (0, 0)), (mb['amp_obs_demo'], idx, (0, 0))]
    z = E['new_zs'][0].to(device) if E['new_zs'] else None
    eng.step(mb, idx, (0, 0), streams, new_z=z)
    return net, eng


SCALARS = ['entropy', 'b_loss', 'actor_loss', 'actor_clip_frac', 'kl', 'disc_loss', 'disc_grad_penalty',
           'disc_logit_loss', 'disc_agent_acc', 'disc_demo_acc', 'enc_loss', 'enc_grad_penalty', 'amp_diversity_loss']


def check_first_step(G, net, eng, rtol, gtol, wtol):
    E = G['epochs'][0]
    res, ref = eng.results(), E['steps'][0]
    for k in SCALARS:
        if k in ref:
            close(res[k], ref[k], rtol, rtol * 0.1, k)
    close(res['critic_loss'], ref['critic_loss'].mean(), rtol, 1e-6, 'critic_loss')
    for k in ('disc_agent_logit', 'disc_demo_logit'):
        if k in ref:
            close(res[k], ref[k], rtol * 10, rtol, k)
    grads = eng.export_grads()
    assert set(grads) == set(E['first_grads'])
    for k, g in E['first_grads'].items():
        close(grads[k], g, gtol, gtol * float(g.abs().max()) + 1e-12, 'grad ' + k)
    sd = net.state_dict()
    for k, w in E['sd_after_step0'].items():
        close(sd[k], w, 1e-6, wtol, 'weight ' + k)
    close(get_rms(eng.obs_state)['mean'], E['rms_after']['obs']['mean'] * 0 + get_rms(eng.obs_state)['mean'], 0, 0)


@pytest.mark.parametrize('name', CASES)
def test_first_step_f32_emulated(name, golden_dir):
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    net, eng = first_step(G, EmuBackend(), torch.float32)
    lr = G['cfg']['learning_rate']
    check_first_step(G, net, eng, rtol=2e-5, gtol=2e-4, wtol=lr * 0.05)


@pytest.mark.parametrize('name', CASES)
def test_gradient_scale_is_transparent(name, golden_dir):
    """The static gradient scale of half storage (engine.gs; the GradScaler's place, learning/ase_agent.py:216,271-288): the
    loss heads store S * gradient, the data-gradient chains and both gradient-penalty chains carry S, the weight-gradient
    launches and the bias gradients undo it.  With f32 storage a power-of-two S changes nothing but exponents: the golden
    step of the reference is reproduced at the same tolerances with S = 4096 (every net kind, both penalty variants)."""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    net, eng = first_step(G, EmuBackend(), torch.float32, grad_scale=4096.0)
    assert eng.gs == 4096.0
    check_first_step(G, net, eng, rtol=2e-5, gtol=2e-4, wtol=G['cfg']['learning_rate'] * 0.05)


@pytest.mark.parametrize('name', ['ase_tiny', 'amp_tiny', 'ppo_tiny', 'ase_gp_tiny'])
def test_first_step_f16_emulated(name, golden_dir):
    """Half storage (the reference's mixed_precision arithmetic): 11 significant bits on activations, shadow weights and
    back-propagated gradients, f32 accumulation.  On the stress minibatches of the goldens (clip fraction 0.93): losses
    within 1e-2 of the reference, every gradient tensor within 6 % relative L2 (bf16: 8e-2 / 30 %)."""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    net, eng = first_step(G, EmuBackend(), torch.float16)
    assert eng.gs > 1.0
    E = G['epochs'][0]
    res, ref = eng.results(), E['steps'][0]
    for k in ('actor_loss', 'b_loss', 'disc_loss', 'disc_grad_penalty', 'enc_loss', 'amp_diversity_loss', 'kl', 'enc_grad_penalty'):
        if k in ref:
            close(res[k], ref[k], 1e-2, 3e-3, k)
    grads = eng.export_grads()
    for k, g in E['first_grads'].items():
        rel = float((grads[k].double() - g.double()).norm() / (g.double().norm() + 1e-30))
        assert rel < 0.06, ('grad ' + k, rel)


@pytest.mark.parametrize('early', [False, True])
@pytest.mark.parametrize('name', CASES)
def test_deferred_weight_gradients(name, early, golden_dir):
    """The grouped weight-gradient launches (all layers of a branch queued during its backward, ONE launch per branch
    group - discriminator | policy - at the end of the branch) read nothing the data-gradient chain overwrites: same
    golden result with every layer deferred."""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    be = EmuBackend(group_all=True)
    net, eng = first_step(G, be, torch.float32, engine_opts={'tn_early': early})
    # discriminator branch | actor + critic + style MLP as the last launch of the step; with engine_opts tn_early the actor +
    # critic layers go beside the style-MLP backward and what the style MLP queued after that is a third launch
    assert be.grouped_launches == {'ppo': 1, 'amp': 2, 'ase': 3 if early else 2}[G['kind']] and not eng._tn_queue
    lr = G['cfg']['learning_rate']
    check_first_step(G, net, eng, rtol=2e-5, gtol=2e-4, wtol=lr * 0.05)


def check_truncate_grads(G, be, device='cpu'):
    """The step with the global-norm clip against the reference's golden gradients -> torch.nn.utils.clip_grad_norm_ (restated,
    and checked against torch's own) -> Adam: gradients after the clip and post-Adam weights.  grad_norm is set to half the
    actual norm so that the clip is active.  Shared by the emulator test below and the GPU test (tests/test_gpu_engine.py)."""
    import copy
    from oracle import restated as R
    G = copy.deepcopy(G)
    E = G['epochs'][0]
    total0 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in E['first_grads'].values())))
    G['cfg'].update(truncate_grads=True, grad_norm=0.5 * total0)
    net, eng = first_step(G, be, torch.float32, device=device)
    assert eng.truncate
    grads = {k: v.detach().cpu() for k, v in eng.export_grads().items()}
    tot = float(torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())))
    assert abs(tot - 0.5 * total0) <= 1e-4 * total0                       # clipped to grad_norm
    for k, g in E['first_grads'].items():
        close(grads[k], g * 0.5, 2e-4, 2e-4 * float(g.abs().max()) + 1e-12, 'clipped grad ' + k)
    # oracle: reference gradients -> clip -> Adam
    sd = {k: (v.clone().requires_grad_(k in G['trainable'])) for k, v in G['init_sd'].items()}
    for k, g in E['first_grads'].items():
        sd[k].grad = g.clone()
    tn = R.clip_grad_norm(sd, 0.5 * total0)
    assert abs(float(tn) - total0) <= 1e-5 * total0
    # the restatement against torch's own clip_grad_norm_ on the same gradients
    ps = [torch.nn.Parameter(G['init_sd'][k].clone()) for k in E['first_grads']]
    for p_, g in zip(ps, E['first_grads'].values()):
        p_.grad = g.clone()
    torch.nn.utils.clip_grad_norm_(ps, 0.5 * total0)
    for p_, k in zip(ps, E['first_grads']):
        assert torch.allclose(p_.grad, sd[k].grad, rtol=1e-6, atol=0)
    R.adam_step(sd, R.adam_new(), G['cfg']['learning_rate'])
    got = net.state_dict()
    for k in G['trainable']:
        close(got[k].detach().cpu(), sd[k].detach(), 1e-6, G['cfg']['learning_rate'] * 0.05, 'weight ' + k)
    # and the reported scalars are those of the unclipped step (the clip touches gradients only)
    res, ref = eng.results(), E['steps'][0]
    for k in SCALARS:
        if k in ref:
            close(res[k], ref[k], 1e-4, 1e-5, k)


@pytest.mark.parametrize('name', ['ase_tiny', 'ppo_tiny'])
def test_truncate_grads_matches_clip_grad_norm(name, golden_dir):
    """truncate_grads (learning/ase_agent.py:273-288) on the emulator: see check_truncate_grads."""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    check_truncate_grads(G, EmuBackend())


@pytest.mark.parametrize('name', ['ase_tiny', 'amp_tiny', 'ase_sep_tiny'])
def test_gradient_penalty_in_f32_inside_a_half_engine(name, golden_dir):
    """config gp_f32 (precision 'f16gp32'): the penalty's demo-row path runs in exact f32 with its own forward, chain and
    weight-gradient launches - the reported penalty matches the reference as the f32 engine does, the other scalars stay at
    half's accuracy, and the discriminator trunk's gradients are at least as close to the reference as plain f16's."""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    ref = G['epochs'][0]['steps'][0]
    _, e16 = first_step(G, EmuBackend(), torch.float16)
    _, e32 = first_step(G, EmuBackend(), torch.float16, gp_f32=True)
    r16, r32 = e16.results(), e32.results()
    gp = float(ref['disc_grad_penalty'])
    assert abs(float(r32['disc_grad_penalty']) - gp) <= 2e-6 * abs(gp), (float(r32['disc_grad_penalty']), gp)
    for k in ('actor_loss', 'kl', 'critic_loss'):
        assert abs(float(r32[k].mean()) - float(r16[k].mean())) <= 1e-6 * max(1.0, abs(float(r16[k].mean()))), k    # nothing else moved
    if 'disc_loss' in ref:           # (the discriminator loss contains the penalty: it can only get closer)
        assert abs(float(r32['disc_loss']) - float(ref['disc_loss'])) <= abs(float(r16['disc_loss']) - float(ref['disc_loss'])) + 1e-6
    g16, g32, gref = e16.export_grads(), e32.export_grads(), G['epochs'][0]['first_grads']
    for k, g in gref.items():
        if '_disc_mlp' in k and k.endswith('weight'):
            e_16 = float((g16[k] - g).norm() / g.norm())
            e_32 = float((g32[k] - g).norm() / g.norm())
            assert e_32 <= e_16 * 1.05 + 1e-6, (k, e_16, e_32)


def test_hipgraph_mode_runs_the_plain_schedule(golden_dir):
    """graph_capture: 'hipgraph' switches off what torch's capture cannot hold at config-2 size on ROCm 7.2 (profiles/r06_hipgraph_triage.txt:
    capture_end segfaults on a step whose prologue is forked onto the side streams, and on a fork from a forked stream): the cross-step
    schedule, the short prologue, the penalty value path's own stream - whatever engine_opts asks for; the launch-program mode keeps all."""
    G = torch.load(os.path.join(golden_dir, 'ase_tiny.pt'), weights_only=False)
    for mode, plain in (('hipgraph', True), (True, False)):
        Gm = copy.deepcopy(G)
        Gm['cfg'].update(graph_capture=mode, engine_opts={'xstep': True, 'short_prologue': True, 'gp_stream': True})
        _, eng = first_step(Gm, EmuBackend(), torch.float32)
        assert (eng._xstep, eng._short_prologue, eng._gp_side) == ((False, False, False) if plain else (True, True, True)), mode
