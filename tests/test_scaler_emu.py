"""The dynamic loss scale of the half-storage engine (config loss_scale: 'dynamic' = what torch.cuda.amp.GradScaler does in the
reference's mixed_precision path, learning/ase_agent.py:271-288): overflow detection over the scaled backward, the skipped
optimizer step, backoff / growth of the DEVICE-resident scale after every optimisation step (ABI 6; round 5 moved the scale on the
host, once per update).  Host logic + op semantics on the CPU emulator; the scale trajectory of the per-step rule is pinned
against torch's own GradScaler over two 48-step updates.  check_dynamic_loss_scale is shared with the GPU test
(tests/test_gpu_scaler.py)."""
import copy
import math
import os

import pytest
import torch

from ase_amd import lib as L
from tests.emu_backend import EmuBackend
from tests.helpers import close, set_rms
from tests.test_engine_emu import first_step


def _step_inputs(G, device):
    E, kind, cfg = G['epochs'][0], G['kind'], G['cfg']
    mb = {k: v.to(device) for k, v in E['first_minibatch'].items()}
    M = mb['obs'].shape[0]
    idx = torch.arange(M, dtype=torch.int32, device=device)
    streams = None
    if kind != 'ppo':
        streams = [(mb['amp_obs'], idx, (0, 0)), (mb['amp_obs_replay'], idx, (0, 0)), (mb['amp_obs_demo'], idx, (0, 0))]
    z = E['new_zs'][0].to(device) if E['new_zs'] else None
    return mb, idx, streams, z


def _reset_rms(G, eng):
    E = G['epochs'][0]
    set_rms(eng.obs_state, E['rms_step0_before']['obs'])
    if G['kind'] != 'ppo':
        set_rms(eng.amp_state, E['rms_step0_before']['amp'])


def _scale_of(eng):
    tab = eng.scale_tab.tolist()
    s = float(eng.scaler[4])
    # the table of scale records {factor, overflow count} follows the state; the counts are consumed by the step
    assert tab[0] == s and abs(tab[2] * s - 1.0) < 1e-6 and abs(tab[4] * s * s - 1.0) < 1e-6 and tab[6] == 1.0, (tab, s)
    assert tab[1::2] == [0.0, 0.0, 0.0, 0.0], tab
    return s


def check_dynamic_loss_scale(G, make_backend, device='cpu', gp_f32=False):
    """One minibatch of a golden, two calls of the dynamic engine against the static-scale engine:
      1. at a scale 2^30 above the static choice the scaled backward saturates half storage: the step is SKIPPED - weights, Adam
         moments and the optimizer's step counter are bit-for-bit what they were, the step's gradient is dropped, the loss scalars
         of the forward (formed in f32) are those of the static engine - and the scale has ALREADY backed off on the device
         (here in one move, backoff_factor 2^-30): no host decision, nothing to re-record;
      2. the same minibatch again, same engine object, same launches, is a CLEAN step at the new scale that equals the static engine's
         step: gradients, post-Adam weights, step counter 1; growth_interval = 1: the scale has grown behind it and the tracker
         starts over."""
    f16 = torch.float16
    lr = G['cfg']['learning_rate']
    sync = torch.cuda.synchronize if str(device) != 'cpu' else (lambda: None)
    # the static-scale engine: the reference point
    net_s, eng_s = first_step(G, make_backend(), f16, device=device, gp_f32=gp_f32)
    sync()
    assert not eng_s.dyn_scale and eng_s._dS is None
    S0 = eng_s.gs
    init_sd = {k: v.detach().clone() for k, v in first_step.__globals__['build_net'](G, device).state_dict().items()}
    Gd = copy.deepcopy(G)
    Gd['cfg'].update(loss_scale='dynamic', loss_scaler={'init_scale': S0 * 2.0 ** 30, 'backoff_factor': 2.0 ** -30,
                                                        'growth_factor': 2.0, 'growth_interval': 1})
    net_d, eng_d = first_step(Gd, make_backend(), f16, device=device, gp_f32=gp_f32)
    sync()
    assert eng_d.dyn_scale and eng_d.gs == 1.0                             # (the host factor: the scale itself lives on the device)
    # ---- 1. the skipped step, and the backoff behind it
    st = eng_d.scaler_state()
    assert (st['skipped'], st['clean'], st['steps']) == (1, 0, 1), st
    assert _scale_of(eng_d) == S0 and st['scale'] == S0
    assert float(eng_d.scaler[0]) == 0.0                                   # the flag is consumed
    assert float(eng_d.opt_state[0]) == 0.0                                # no optimizer step happened
    sd = net_d.state_dict()
    for k, v in init_sd.items():
        assert torch.equal(sd[k].detach().cpu(), v.cpu()), 'weight moved in a skipped step: ' + k
    assert float(eng_d.adam_m.abs().max()) == 0.0 and float(eng_d.adam_v.abs().max()) == 0.0
    # the step's gradient was dropped: what the exported buffer holds afterwards is at most the weight-only loss terms c * W the fused
    # optimizer launch adds behind the decision (the identity step ignores them) - nothing of the overflowed backward, nothing non-finite
    gmax = float(eng_d.grads[:eng_d.n_train].abs().max())
    assert math.isfinite(gmax) and gmax <= 0.25 * float(eng_d.params[:eng_d.n_train].abs().max()) + 1e-12, gmax
    rs, rd = eng_s.results(), eng_d.results()
    for k in ('actor_loss', 'critic_loss', 'kl'):        # (not the penalties: their values come out of the scaled chains)
        if k in rs:
            close(rd[k], rs[k], 1e-5, 1e-6, k + ' (skipped step)')
    assert eng_d.scaler_update() is False                                  # (round-5 interface: nothing left for the host)
    # ---- 2. the same minibatch as a clean step at the backed-off scale
    _reset_rms(G, eng_d)
    mb, idx, streams, z = _step_inputs(G, device)
    eng_d.step(mb, idx, (0, 0), streams, new_z=z)
    sync()
    st = eng_d.scaler_state()
    assert (st['skipped'], st['clean'], st['steps']) == (1, 0, 2), st       # clean step -> growth (interval 1) -> tracker starts over
    assert _scale_of(eng_d) == 2.0 * S0
    assert float(eng_d.opt_state[0]) == 1.0
    gs_, gd_ = eng_s.export_grads(), eng_d.export_grads()
    for k, g in gs_.items():
        close(gd_[k], g, 2e-5, 2e-5 * float(g.abs().max()) + 1e-12, 'grad ' + k)
    sd, ss = net_d.state_dict(), net_s.state_dict()
    moved = 0.0
    for k in G['trainable']:
        close(sd[k], ss[k], 1e-6, lr * 0.25, 'weight ' + k)     # (Adam's first step is +-lr whatever |g|: sign noise of g ~ eps)
        moved = max(moved, float((sd[k].detach().cpu() - init_sd[k].cpu()).abs().max()))
    assert moved > 0.5 * lr                                                # (Adam's first step moves every weight by ~lr)
    rd = eng_d.results()
    for k in ('actor_loss', 'critic_loss', 'kl', 'disc_loss', 'disc_grad_penalty', 'enc_loss'):
        if k in rs:
            close(rd[k], rs[k], 1e-5, 1e-6, k + ' (clean step)')
    return eng_d


@pytest.mark.parametrize('gp_f32', [False, True])
@pytest.mark.parametrize('name', ['ase_tiny', 'amp_tiny', 'ppo_tiny', 'ase_sep_gp_tiny'])
def test_dynamic_loss_scale_skips_and_recovers(name, gp_f32, golden_dir):
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    if gp_f32 and G['kind'] == 'ppo':
        pytest.skip('no discriminator')
    check_dynamic_loss_scale(G, EmuBackend, gp_f32=gp_f32)


def test_scaler_ops_emulated():
    """Op semantics (csrc/scaler.hip): what counts as an overflow per storage type, and the two outcomes of scaler_step."""
    be = EmuBackend()
    sc = torch.zeros(8, dtype=torch.float64)
    ok16 = torch.tensor([1.0, -65472.0, 0.0, 6e-8], dtype=torch.float16)
    be.scaler_check(ok16, sc)
    be.scaler_check(torch.tensor([1.0, 3e38, -3e38]), sc)
    be.scaler_check(torch.tensor([1.0, 3e38], dtype=torch.bfloat16), sc)
    assert float(sc[0]) == 0.0
    for bad in (torch.tensor([0.0, 65504.0], dtype=torch.float16), torch.tensor([-65504.0], dtype=torch.float16),
                torch.tensor([float('nan')], dtype=torch.float16), torch.tensor([float('inf')]),
                torch.tensor([float('nan'), 1.0]), torch.tensor([float('-inf')], dtype=torch.bfloat16)):
        sc.zero_()
        be.scaler_check(bad, sc)
        assert float(sc[0]) > 0.0, bad
    opt = torch.tensor([3.0, 2e-5, 0.9, 0.999, 1e-8, 0.271, 0.003, 0.0], dtype=torch.float64)
    eff = torch.zeros(8, dtype=torch.float64)
    g = torch.ones(10)
    sc.zero_()
    be.scaler_step(sc, opt, eff, g)                                         # clean
    assert torch.equal(eff, opt) and float(g.sum()) == 10.0 and sc.tolist()[:4] == [0.0, 0.0, 1.0, 1.0]
    sc[0] = 2.0
    be.scaler_step(sc, opt, eff, g)                                         # overflow
    assert float(g.abs().sum()) == 0.0 and float(opt[0]) == 2.0
    assert eff.tolist() == [2.0, 0.0, 1.0, 1.0, 1e-8, 1.0, 1.0, 0.0] and sc.tolist()[:4] == [0.0, 1.0, 0.0, 2.0]
    # GradScaler.update() on the device state (scale_tab given): backoff at once, growth when the tracker reaches the interval
    sc = torch.tensor([1.0, 0, 0, 0, 1024.0, 2.0, 0.5, 2.0], dtype=torch.float64)
    tab = torch.zeros(8)
    be.scaler_step(sc, opt, eff, g, scale_tab=tab)                          # overflow: 1024 -> 512
    assert sc.tolist() == [0.0, 1.0, 0.0, 1.0, 512.0, 2.0, 0.5, 2.0]
    assert tab.tolist() == [512.0, 0.0, 1 / 512.0, 0.0, 1 / 512.0 ** 2, 0.0, 1.0, 0.0]
    be.scaler_step(sc, opt, eff, g, scale_tab=tab)                          # clean 1 of 2
    assert sc.tolist()[:5] == [0.0, 1.0, 1.0, 2.0, 512.0]
    be.scaler_step(sc, opt, eff, g, scale_tab=tab)                          # clean 2 of 2: growth, tracker starts over
    assert sc.tolist()[:5] == [0.0, 1.0, 0.0, 3.0, 1024.0] and tab.tolist()[:3] == [1024.0, 0.0, 1 / 1024.0]
    # what the PRODUCERS reported into a record's count (ABI 7) decides like scaler[found]: skipped step, backoff, counts consumed
    g = torch.ones(10)
    tab[5] = 3.0
    be.scaler_step(sc, opt, eff, g, scale_tab=tab)
    assert float(g.abs().sum()) == 0.0 and sc.tolist()[:5] == [0.0, 2.0, 0.0, 4.0, 512.0] and tab.tolist()[1::2] == [0.0] * 4
    # ... and scaler_fold moves the counts into scaler[found] (what a data-parallel step exchanges)
    tab[1], tab[7] = 1.0, 2.0
    be.scaler_fold(sc, tab)
    assert float(sc[0]) == 3.0 and tab.tolist()[1::2] == [0.0] * 4
    sc[0] = 0.0
    # the identity step through the optimizer op: nothing moves, even with moments in place
    opt = torch.tensor([3.0, 2e-5, 0.9, 0.999, 1e-8, 0.271, 0.003, 0.0], dtype=torch.float64)
    sc = torch.zeros(8, dtype=torch.float64)
    sc[0] = 2.0
    be.scaler_step(sc, opt, eff, g)
    w, m, v = torch.randn(10), torch.randn(10) * 1e-3, torch.rand(10) * 1e-6
    w0, m0, v0 = w.clone(), m.clone(), v.clone()
    be.adam(w, g, m, v, eff)
    assert torch.equal(w, w0) and torch.equal(m, m0) and torch.equal(v, v0)


def test_producers_report_into_their_record():
    """A launch that is given a scale record {factor, count} multiplies the factor in and counts an overflow it STORED (csrc/common.h
    ovf_report): matrix launch (output and pre-activation twin), the loss heads' stored gradients; nothing is counted for finite values,
    for what a ReLU clips away, or without a record."""
    be = EmuBackend()
    f16 = torch.float16
    A = torch.full((4, 64), 30.0, dtype=f16)
    B = torch.full((32, 64), 40.0, dtype=f16)            # 64 * 1200 = 76800 > 65504
    C = torch.zeros(4, 32, dtype=f16)
    rec = torch.tensor([1.0, 0.0])
    be.gemm_nt(A, B, C, 4, 32, 64, alpha_dev=rec)
    assert float(rec[1]) > 0 and float(C.float().abs().max()) == 65504.0
    rec = torch.tensor([0.5, 0.0])                        # the factor is applied first: 38400 fits
    be.gemm_nt(A, B, C, 4, 32, 64, alpha_dev=rec)
    assert float(rec[1]) == 0 and float(C[0, 0]) == 38400.0
    rec = torch.tensor([-1.0, 0.0])                       # -76800 -> ReLU stores 0: nothing overflowed in storage
    be.gemm_nt(A, B, C, 4, 32, 64, act=L.ACT_RELU, alpha_dev=rec)
    assert float(rec[1]) == 0 and float(C.abs().max()) == 0.0
    Cb = torch.zeros(4, 32, dtype=torch.bfloat16)         # bf16 storage does not saturate: finite stays finite
    rec = torch.tensor([1.0, 0.0])
    be.gemm_nt(A.to(torch.bfloat16), B.to(torch.bfloat16), Cb, 4, 32, 64, alpha_dev=rec)
    assert float(rec[1]) == 0
    be.gemm_nt(A, B, C, 4, 32, 64)                        # no record: nothing to write to
    logit = torch.zeros(12, 1)
    d = torch.zeros(12, 1, dtype=f16)
    acc = torch.zeros(L.ACC_COUNT, dtype=torch.float64)
    rec = torch.tensor([2.0 ** 40, 0.0])
    be.disc_head(logit, d, None, acc, 4, 4, 5.0, grad_scale=1.0, dyn=rec)
    assert float(rec[1]) > 0
    rec = torch.tensor([2.0 ** 10, 0.0])
    be.disc_head(logit, d, None, acc, 4, 4, 5.0, grad_scale=1.0, dyn=rec)
    assert float(rec[1]) == 0


def test_scale_trajectory_matches_torch_gradscaler(golden_dir):
    """The device-side rule (scaler_step with the engine's state and table) against torch.amp.GradScaler on the same found_inf
    sequence over TWO UPDATES OF 48 OPTIMISATION STEPS: the scale after every step (the reference calls scaler.update() behind every
    scaler.step(), learning/ase_agent.py:280,285,288), the number of optimizer steps taken, the skipped count - one step lost per
    backoff, not the rest of the update (round 5: 36 skipped steps for two backoffs)."""
    G = torch.load(os.path.join(golden_dir, 'ppo_tiny.pt'), weights_only=False)
    Gd = copy.deepcopy(G)
    Gd['cfg'].update(loss_scale='dynamic', loss_scaler={'init_scale': 2.0 ** 16, 'growth_interval': 7})
    _, eng = first_step(Gd, EmuBackend(), torch.float16)
    eng.scaler[:4] = 0.0
    eng.set_grad_scale(2.0 ** 16)
    p = torch.nn.Parameter(torch.zeros(4))
    opt = torch.optim.SGD([p], lr=0.1)
    ref = torch.amp.GradScaler('cpu', init_scale=2.0 ** 16, growth_factor=2.0, backoff_factor=0.5, growth_interval=7)
    g = torch.Generator().manual_seed(11)
    pattern = [1, 1] + [int(x) for x in (torch.rand(94, generator=g) < 0.08)]       # two backoffs at the start (65536 -> 16384), then rare ones
    assert len(pattern) == 2 * 48 and 2 < sum(pattern) < 20
    taken_ref = taken = 0
    for bad in pattern:
        # torch: a scaled backward whose gradient is inf when `bad`
        opt.zero_grad()
        c = torch.full((4,), 3e38 if bad else 1.0)                        # (d loss / d p = scale * c overflows f32 when bad)
        ref.scale((p * c).sum()).backward()
        before = p.detach().clone()
        ref.step(opt)
        ref.update()
        taken_ref += int(not torch.equal(before, p.detach()))
        # here: the device-side decision and update on the same flag
        eng.scaler[0] = float(bad)
        gr = torch.ones(4)
        eng.be.scaler_step(eng.scaler, eng.opt_state, eng.opt_eff, gr, scale_tab=eng.scale_tab)
        taken += int(float(eng.opt_eff[1]) != 0.0)
        assert _scale_of(eng) == ref.get_scale(), (_scale_of(eng), ref.get_scale())
    assert taken == taken_ref == pattern.count(0)
    st = eng.scaler_state()
    assert st['skipped'] == pattern.count(1) and st['steps'] == 96 and st['scale'] == ref.get_scale()


def _agent_without_precision_key(G, device='cpu', backend=None):
    """tests.test_agent_emu.make_agent minus its `precision` key: the configuration a reference yaml with mixed_precision: True gives."""
    import types
    from tests import test_agent_emu as T
    from tests.helpers import BUILDERS, golden_init_sd
    kind, spec = G['kind'], G['spec']
    b = BUILDERS[kind]()
    b.load(G['net'])
    sp = lambda n: types.SimpleNamespace(shape=(n,))
    cfg = dict(G['cfg'])
    cfg.pop('precision', None)
    cfg.update(network=T.MODELS[kind](b), num_actors=spec['num_envs'], device=device, backend=backend or EmuBackend(),
               env_info={'observation_space': sp(spec['obs_size']), 'action_space': sp(spec['act_size']),
                         'amp_observation_space': sp(spec['amp_obs_size'])}, vec_env=T._Feed())
    ag = T.AGENTS[kind]('golden', cfg)
    ag.model.load_state_dict({'a2c_network.' + k: v.to(device) for k, v in golden_init_sd(G).items()})
    ag.engine.refresh_shadows()
    return ag


def test_mixed_precision_flag_selects_the_dynamic_scale(golden_dir):
    """The reference's own flag (mixed_precision: True, no precision key) = half storage WITH the GradScaler's behaviour; a named
    precision mode keeps the static scale; the scale moves per optimisation step on the device and NO recorded launch program is
    dropped for it."""
    from tests.test_agent_emu import make_agent, replay_epochs
    G = torch.load(os.path.join(golden_dir, 'ase_tiny.pt'), weights_only=False)
    ag = make_agent(copy.deepcopy(G), EmuBackend(), precision='f16')
    assert not ag.engine.dyn_scale and ag.engine.gs == 2.0 ** max(0, round(math.log2(max(ag.minibatch_size, 4) / 4.0)))
    Gm = copy.deepcopy(G)
    Gm['cfg'].update(mixed_precision=True, loss_scaler={'init_scale': 2.0 ** 40})
    ag = _agent_without_precision_key(Gm)
    assert ag.precision == 'f16' and ag.engine.dyn_scale and ag.engine.gs == 1.0 and ag.engine.scaler_state()['scale'] == 2.0 ** 40
    dropped = []
    drop = ag._drop_graphs
    ag._drop_graphs = lambda: (dropped.append(1), drop())
    replay_epochs(Gm, ag, rtol=1.0, wtol=1.0, check=False, max_steps=2)
    st = ag.engine.scaler_state()
    n_updates = len(Gm['epochs'])
    assert st['steps'] == 2 * n_updates and st['skipped'] >= 2             # at 2^40 every step overflows: each one skipped, each one halves
    assert st['scale'] == 2.0 ** 40 * 0.5 ** st['skipped']
    assert not dropped


def test_dynamic_loss_scale_with_truncate_grads(golden_dir):
    """scaler.unscale_ -> clip_grad_norm_ -> scaler.step (learning/ase_agent.py:273-285): with the clip active a clean step of the
    dynamic engine is the static engine's clipped step; an overflowing one clips nothing (zero gradient) and moves nothing."""
    G = torch.load(os.path.join(golden_dir, 'ase_tiny.pt'), weights_only=False)
    E = G['epochs'][0]
    total0 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in E['first_grads'].values())))
    Gs = copy.deepcopy(G)
    Gs['cfg'].update(truncate_grads=True, grad_norm=0.5 * total0)
    net_s, eng_s = first_step(Gs, EmuBackend(), torch.float16)
    assert eng_s.truncate and not eng_s.dyn_scale
    for init, skipped in ((eng_s.gs, 0), (eng_s.gs * 2.0 ** 30, 1)):
        Gd = copy.deepcopy(Gs)
        Gd['cfg'].update(loss_scale='dynamic', loss_scaler={'init_scale': init})
        net_d, eng_d = first_step(Gd, EmuBackend(), torch.float16)
        assert eng_d.truncate and eng_d.dyn_scale and eng_d.scaler_state()['skipped'] == skipped
        gd = eng_d.export_grads()
        if skipped:
            assert all(float(v.abs().max()) == 0.0 for v in gd.values()) and float(eng_d.opt_state[0]) == 0.0
            for k, v in G['init_sd'].items():
                assert torch.equal(net_d.state_dict()[k], v), k
        else:
            gs_ = eng_s.export_grads()
            tot = float(torch.sqrt(sum((g.double() ** 2).sum() for g in gd.values())))
            assert abs(tot - 0.5 * total0) <= 2e-2 * total0                  # clipped to grad_norm (half storage: the golden's norm +- 2 %)
            for k, g in gs_.items():
                close(gd[k], g, 2e-5, 2e-5 * float(g.abs().max()) + 1e-12, 'clipped grad ' + k)
            for k in G['trainable']:
                close(net_d.state_dict()[k], net_s.state_dict()[k], 1e-6, G['cfg']['learning_rate'] * 0.25, 'weight ' + k)
