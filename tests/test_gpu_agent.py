"""GPU parity tests, epoch level: the agents' train_epoch update on the HIP backend replayed against the two
full train_epoch calls of the reference recorded in the golden vectors (same inputs, same injected random
draws): per-step losses, final weights after 8 optimisation steps per epoch, running statistics, replay ring."""
import os

import pytest
import torch

from tests.test_agent_emu import REAL_WIDTH, check_real_width, make_agent, replay_epochs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
    from ase_amd.backend import HipBackend
    return HipBackend()


@pytest.mark.parametrize('name', ['ase_tiny', 'amp_tiny', 'ppo_tiny', 'ase_sep_tiny', 'ase_tiny_s1', 'ase_tiny_s2', 'amp_cfg1', 'ase_gp_tiny',
                                  'ase_sep_gp_tiny', 'ase_swish_tiny'])
def test_two_epochs_f32(be, name, golden_dir):
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    ag = make_agent(G, be, device='cuda', precision='f32')
    replay_epochs(G, ag, rtol=3e-4, wtol=G['cfg']['learning_rate'] * 0.25)


@pytest.mark.parametrize('name', REAL_WIDTH)
def test_real_width_reference_goldens_f32(be, name, golden_dir):
    """The UNMODIFIED reference at the real layer widths (ase_humanoid.yaml / hrl_humanoid.yaml nets verbatim; rows a9 / a22):
    first-step losses + sampled gradients / post-Adam weights at BASELINE's 1e-4, then all 8 steps of the update."""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    lr = float(G['cfg']['learning_rate'])        # (PyYAML reads the yaml's '2e-5' as a string)
    check_real_width(G, lambda: make_agent(G, be, device='cuda', precision='f32'), rtol=1e-4, gtol=3e-4, wtol=lr * 0.25,
                     traj_rtol=3e-4)


@pytest.mark.parametrize('precision,loss_tol', [('f16', 2e-3), ('bf16', 2e-2)])
@pytest.mark.parametrize('name', ['ase_cfg2_small', 'hrl_cfg4_small'])
def test_real_width_reference_goldens_16bit(be, name, precision, loss_tol, golden_dir):
    """The 16-bit storage modes against the same reference goldens (foreign old log-probabilities: the reference's f32
    rollout feeds the 16-bit update, so an error d in mu is amplified by |a - mu| / sigma^2 - the hard case; bench.py
    measures the self-consistent one): every step's loss scalars within the stated bound relative to their scale."""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    ag = make_agent(G, be, device='cuda', precision=precision)
    infos = replay_epochs(G, ag, rtol=0, wtol=0, check=False)
    E = G['epochs'][0]
    scale = {'actor_loss': 1.0, 'enc_loss': 1.0, 'kl': 0.1, 'b_loss': 1.0}
    for i, ref in enumerate(E['steps']):
        # (kl is left out: it is quadratic in (mu_new - mu_old) / sigma, and here mu_old is the reference's f32 rollout)
        for k in ('actor_loss', 'critic_loss', 'b_loss', 'disc_loss', 'disc_grad_penalty', 'enc_loss', 'amp_diversity_loss'):
            if k in ref:
                a, b = float(infos[0][k][i]), float(ref[k].mean())
                if abs(b) > 1e3:        # importance ratios of e^20+ (the HRL golden: 64 action dims, untrained statistics):
                    continue            # the reference's own value is an overflowing sum, not a target
                assert a == a and abs(a - b) <= loss_tol * max(abs(b), scale.get(k, 0.0)), (precision, k, i, a, b)


def _first_step_errors(G, ag):
    """(|delta| / max(|ref|, floor), |delta| / |ref|) per loss scalar of the FIRST optimisation step against the reference's
    recorded train_result; floor = 1 for the two means of signed O(1) summands (actor_loss, enc_loss), 0 otherwise."""
    import copy
    infos = replay_epochs(copy.deepcopy(G), ag, rtol=0, wtol=0, check=False, max_steps=1)
    ref = G['epochs'][0]['steps'][0]
    floor = {'actor_loss': 1.0, 'enc_loss': 1.0}
    out = {}
    for k in ('actor_loss', 'critic_loss', 'b_loss', 'kl', 'entropy', 'disc_loss', 'disc_grad_penalty', 'disc_logit_loss', 'enc_loss',
              'amp_diversity_loss'):
        if k in ref:
            a, b = float(infos[0][k][0]), float(ref[k].mean())
            assert a == a, k
            out[k] = (abs(a - b) / max(abs(b), floor.get(k, 0.0), 1e-12), abs(a - b) / max(abs(b), 1e-12))
    return out


def _post_adam_and_gradients(G, ag):
    """Gradients and post-Adam weights of the FIRST optimisation step of a half-storage engine against the reference's goldens at the
    real widths (round 4's verdict: only f32 was checked).  Gradients: relative L2 on the goldens' seeded samples - bounded by
    what ReLU mask flips leave (profiles/r05_grad_error_sources.txt: a unit whose pre-activation changes sign under the forward's
    2^-12 rounding is an O(1) error of one element of dZ; a flipped fraction f is sqrt(f) in relative L2, ~1-3 % in half, whatever
    the backward's precision).  Weights: the first Adam step moves every element by exactly +-lr (m / sqrt(v) = sign(g)), so an
    element differs by 2 lr where the two gradients disagree in SIGN and by ~0 elsewhere: no element beyond 2.2 lr, at most 6 % of
    the sampled elements beyond lr / 2, mean difference below 0.12 lr."""
    from tests.helpers import sample_index
    E, sseed = G['epochs'][0], G['sample']['seed']
    lr = float(G['cfg']['learning_rate'])
    grads = ag.engine.export_grads()
    rels = {}
    for k, g in E['first_grads'].items():
        a = grads[k].detach().cpu().reshape(-1)
        ref = g['vals'] if isinstance(g, dict) else g.reshape(-1)
        if isinstance(g, dict):
            a = a[sample_index(k, a.numel(), ref.numel(), sseed)]
        rels[k] = float((a.double() - ref.double()).norm() / ref.double().norm().clamp_min(1e-30))
    vals = sorted(rels.values())
    print('first-step gradient rel L2 vs the reference: median', vals[len(vals) // 2], 'worst', max(rels, key=rels.get), vals[-1])
    assert vals[len(vals) // 2] <= 0.05 and vals[-1] <= 0.15, rels
    sd = ag.model.state_dict()
    for k, w in E['sd_after_step0'].items():
        a = sd['a2c_network.' + k].detach().cpu().reshape(-1)
        ref = w['vals'] if isinstance(w, dict) else w.reshape(-1)
        if isinstance(w, dict):
            a = a[sample_index(k, a.numel(), ref.numel(), sseed)]
        e = (a.double() - ref.double()).abs()
        assert float(e.max()) <= 2.2 * lr + 1e-6 * float(ref.abs().max()), ('weight after step 0 ' + k, float(e.max()) / lr)
        assert float((e > 0.5 * lr).double().mean()) <= 0.06 and float(e.mean()) <= 0.12 * lr, \
            ('weight after step 0 ' + k, float((e > 0.5 * lr).double().mean()), float(e.mean()) / lr)


@pytest.mark.parametrize('precision', ['f16gpx3', 'f16gp32'])
@pytest.mark.parametrize('name', ['ase_cfg2_small', 'ase_cfg2_small_s1', 'ase_cfg2_small_s2'])
def test_real_width_reference_goldens_qualifying_modes(be, name, precision, golden_dir):
    """The modes the bench names as qualifying, against the UNMODIFIED reference's first optimisation step at the real layer
    widths (learning/ase_agent.py:228-258, learning/amp_agent.py:442-479).  Everything that does not pass through the foreign
    old log-probabilities holds BASELINE's 1e-4: the gradient penalty (what the mode exists for) 1e-5 / 5e-5, discriminator /
    critic / encoder / diversity losses 1e-4, kl 2e-4 of its own value.  actor_loss alone carries the foreign-rollout
    amplification (the reference's f32 mu_old against this engine's f16 mu: d logp = (a - mu) / sigma^2 d mu ~ 18 / sigma d mu):
    5e-4 of its summand scale.  (Emulator, same goldens: 8e-8 / 1e-5 / 4e-5 / 4e-5 ... 1.5e-4.)"""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    ag = make_agent(G, be, device='cuda', precision=precision)
    err = _first_step_errors(G, ag)
    # (exact-f32 value path: 1e-5; three f16 MFMAs on hi / lo splits of scaled operands, unit roundoff ~2^-22 per product: 1e-5 as
    #  well - round 4's bf16 split, ~2^-17, was held to 5e-5 here and measured 2.2e-5)
    assert err['disc_grad_penalty'][1] <= 1e-5, err
    _post_adam_and_gradients(G, ag)
    for k in ('critic_loss', 'disc_loss', 'disc_logit_loss', 'enc_loss', 'amp_diversity_loss', 'entropy', 'b_loss'):
        if k in err:
            assert err[k][0] <= 1e-4, (k, err)
    assert err['kl'][1] <= 2e-4, err
    assert err['actor_loss'][0] <= 5e-4, err


@pytest.mark.parametrize('act', ['elu', 'gelu', 'softplus', 'selu', 'sigmoid'])
def test_activation_family_f32(be, act, golden_dir):
    """The reference agent with `activation: <act>` in every MLP (oracle/make_golden.py acts; emulator twin:
    tests/test_agent_emu.py::test_activation_family_against_the_reference): a whole update on the GPU in f32, incl. the
    gradient penalty's double backward through the curved activation.  (First run on hardware: round 3's driver, green.)"""
    G = torch.load(os.path.join(golden_dir, f'ase_{act}_tiny.pt'), weights_only=False)
    ag = make_agent(G, be, device='cuda', precision='f32')
    replay_epochs(G, ag, rtol=3e-4, wtol=float(G['cfg']['learning_rate']) * 0.25)


@pytest.mark.parametrize('name', ['ase_swish_tiny', 'ase_sigmoid_tiny'])
def test_smooth_activation_discriminator_f16(be, name, golden_dir):
    """Half storage through the CURVED gradient penalty (UpdateEngine._disc_backward_curved: the Sc / Sr split of the gradient
    scale, pre-activation twins in f16, ase_hip_gp_second's act'' / act' path): first step of the reference's swish / sigmoid
    agents - penalty within 5e-4 of its value, the other losses within 2e-3 of their scale, everything finite over the
    whole two-epoch replay.  (Emulator: 6e-6 / 1.2e-4 on the penalty.)"""
    import copy
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    err = _first_step_errors(G, make_agent(G, be, device='cuda', precision='f16'))
    assert err['disc_grad_penalty'][1] <= 5e-4, err
    for k, (e, _) in err.items():
        assert e <= 2e-3, (k, err)
    ag = make_agent(G, be, device='cuda', precision='f16')
    infos = replay_epochs(copy.deepcopy(G), ag, rtol=0, wtol=0, check=False)
    for info in infos:
        for k, v in info.items():
            for x in v:
                if torch.is_tensor(x):
                    assert bool(torch.isfinite(x.float()).all()), k
    assert bool(torch.isfinite(ag.model.a2c_network.flat_params).all())


@pytest.mark.parametrize('name', ['amp_tiny', 'ppo_tiny'])
def test_two_epochs_f32_hipgraph(be, name, golden_dir):
    """Same, with every optimisation step replayed from a captured hipGraph (no injected latents needed)."""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    ag = make_agent(G, be, device='cuda', precision='f32', graph_capture=True)
    replay_epochs(G, ag, rtol=3e-4, wtol=G['cfg']['learning_rate'] * 0.25)
    assert len(ag._graphs) >= 1


@pytest.mark.parametrize('name', ['ase_tiny', 'amp_tiny'])
def test_two_epochs_bf16_runs(be, name, golden_dir):
    """bf16 mode end to end: finite, and the losses of every step within 10% (+0.05) of the reference's."""
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    ag = make_agent(G, be, device='cuda', precision='bf16')
    infos = replay_epochs(G, ag, rtol=0, wtol=0, check=False)
    for info, E in zip(infos, G['epochs']):
        for i, ref in enumerate(E['steps']):
            for k in ('disc_loss', 'enc_loss', 'critic_loss', 'amp_diversity_loss'):
                if k in ref:
                    a, b = float(info[k][i]), float(ref[k].mean())
                    assert a == a and abs(a - b) <= 0.1 * abs(b) + 0.05, (k, i, a, b)


def test_graph_and_eager_agree_ase(be, golden_dir):
    """ASE step under hipGraph replay (latents drawn on device) == eager step with the same latents injected."""
    G = torch.load(os.path.join(golden_dir, 'ase_tiny.pt'), weights_only=False)
    res = []
    for graph in (False, True):
        ag = make_agent(G, be, device='cuda', precision='f32', graph_capture=graph)
        ag.engine.div_rng[0] = 777
        E = G['epochs'][0]
        ag.vec_env.q.append(G['demo_init'].clone())
        for k, v in E['exp'].items():
            if k in ag.experience:
                ag.experience[k].copy_(v)
        batch = ag._play_steps_tail()
        ag.vec_env.q.append(E['demo_fetched'].clone())
        ag._amp_obs_demo_buffer._sample_idx = E['demo_sample_perm'].cuda()
        info = ag.update(batch, perms=E['dataset_perms'])
        res.append(({k: torch.stack([x.float() for x in v]).cpu() for k, v in info.items() if torch.is_tensor(v[0]) and v[0].numel() == 1},
                    ag.model.a2c_network.flat_params.clone().cpu()))
    for k in res[0][0]:
        assert torch.allclose(res[0][0][k], res[1][0][k], rtol=1e-4, atol=1e-6), k
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=2e-6)


def _dp_worker(rank, world, port, name, precision, graph):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ase_amd.backend import HipBackend
    G = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.pt'), weights_only=False)
    ag = make_agent(G, HipBackend('cuda:0'), device='cuda:0', precision=precision, world_size=world, rank=rank,
                    graph_capture=graph)
    replay_epochs(G, ag, rtol=3e-4, wtol=G['cfg']['learning_rate'] * 0.25)       # against the REFERENCE's run
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('name,graph', [('ase_tiny', False), ('amp_tiny', True)])
def test_two_ranks_sharded_update_matches_reference(name, graph):
    """Data-parallel HIP path: two ranks (sharing this box's single GPU; collectives staged through gloo) shard every
    minibatch, all-reduce normaliser moments, mask sums and the flat gradient buffer, and must reproduce the
    reference's two train_epochs.  With graph=True each rank replays the three per-phase hipGraphs."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_dp_worker, args=(2, port, name, 'f32', graph), nprocs=2, join=True)


def _dp_scaler_worker(rank, world, port, name, out):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from ase_amd.backend import HipBackend
    G = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', name + '.pt'), weights_only=False)
    ag = make_agent(G, HipBackend('cuda:0'), device='cuda:0', precision='f16', world_size=world, rank=rank, graph_capture=True,
                    loss_scale='dynamic', loss_scaler={'init_scale': 16.0})
    eng = ag.engine
    assert eng.dyn_scale and eng.scaler_state()['scale'] == 16.0
    if rank == 1:            # an overflow only THIS rank's launches report, in the very first step: a count in its record of factor 1
        eng.scale_tab[7] = 1.0
    replay_epochs(G, ag, rtol=1.0, wtol=1.0, check=False, max_steps=2)
    torch.cuda.synchronize()
    flat = ag.model.a2c_network.flat_params.detach().cpu().clone()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    states = [None] * world
    dist.all_gather_object(states, eng.scaler_state())
    if rank == 0:
        torch.save({'flats': gathered, 'states': states, 'opt_step': float(eng.opt_state[0]), 'xs': bool(eng._xs)}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_skip_the_same_step(tmp_path, golden_dir):
    """The dynamic loss scale under data parallelism on the HIP path, inside recorded launch programs: the producers' reports are folded
    (ase_hip_scaler_fold), the flag is SUM-exchanged as a host-callback entry of the step's program in front of the ONE decision
    (UpdateEngine._dyn_apply, on the discriminator's stream) - a step only rank 1 saw an overflow in is skipped by BOTH ranks, the
    replicas stay bit-identical, the scale backs off once on both.  (CPU counterpart: tests/test_dp_gloo.py::test_two_ranks_skip_the_same_step.)"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'sc.pt')
    mp.spawn(_dp_scaler_worker, args=(2, port, 'ase_tiny', out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    assert torch.equal(r['flats'][0], r['flats'][1])
    s0, s1 = r['states']
    assert s0 == s1, (s0, s1)
    G = torch.load(os.path.join(golden_dir, 'ase_tiny.pt'), weights_only=False)
    n_upd = len(G['epochs'])
    assert s0['steps'] == 2 * n_upd and s0['skipped'] == 1 and s0['scale'] == 8.0, s0
    assert r['opt_step'] == 2 * n_upd - 1


@pytest.mark.parametrize('name', ['ase_tiny', 'amp_tiny', 'ppo_tiny'])
@pytest.mark.parametrize('precision', ['f32', 'bf16'])
def test_rollout_inference_matches_reference(be, name, precision, golden_dir):
    """N1: get_action_values / _eval_critic on the HIP inference path against the reference's recorded rollout."""
    from tests.test_agent_emu import check_rollout_inference
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    ag = make_agent(G, be, device='cuda', precision=precision)
    if precision == 'f32':
        check_rollout_inference(G, ag, rtol=2e-5, atol=2e-6)
    else:
        check_rollout_inference(G, ag, rtol=3e-2, atol=1e-2)


@pytest.mark.parametrize('name', ['ase_tiny', 'ppo_tiny'])
def test_checkpoint_interop_with_reference_gpu(be, name, golden_dir):
    """N3 on the HIP path: our checkpoint after the two epochs == the reference's; restore() of the reference's."""
    from tests.test_agent_emu import check_checkpoint_interop
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    ag = make_agent(G, be, device='cuda', precision='f32')
    replay_epochs(G, ag, rtol=3e-4, wtol=G['cfg']['learning_rate'] * 0.25, check=False)
    check_checkpoint_interop(G, ag, lambda: make_agent(G, be, device='cuda', precision='f32'),
                             wtol=G['cfg']['learning_rate'] * 0.25)
