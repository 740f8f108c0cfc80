"""torch.ops.ase_hip.* (ase_amd/ops.py: the PyTorch custom-operator layer over the C ABI) against plain PyTorch on the same
device: forward values, autograd through ase_hip::linear_act (an nn.Module built from HipLinear trains with loss.backward()
+ torch.optim.Adam like the reference's networks), the normaliser's running statistics, GAE, masked advantage normalisation."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _ops():
    import ase_amd.ops as O
    return O


@pytest.mark.parametrize('dt,tol', [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize('act', ['none', 'relu', 'tanh'])
def test_linear_act_forward_backward(dt, tol, act):
    O = _ops()
    g = torch.Generator().manual_seed(0)
    M, K, N = 300, 253, 70                    # ragged on purpose: the operator pads to the kernels' granules
    x = torch.randn(M, K, generator=g).to(DEV).to(dt).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) * 0.1).to(DEV).requires_grad_(True)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV).requires_grad_(True)
    y = torch.ops.ase_hip.linear_act(x, w, b, act)
    xr, wr, br = x.detach().float().requires_grad_(True), w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    wq = wr.to(dt).float() if dt == torch.bfloat16 else wr
    pre = xr @ wq.t() + br
    yr = {'none': pre, 'relu': torch.relu(pre), 'tanh': torch.tanh(pre)}[act]
    assert y.shape == (M, N) and y.dtype == dt
    scale = float(yr.detach().abs().max())
    assert float((y.detach().float() - yr.detach()).abs().max()) <= tol * scale
    dy = torch.randn(M, N, generator=g).to(DEV)
    y.backward(dy.to(dt))
    yr.backward(dy)
    for a, r, name in ((x.grad, xr.grad, 'dx'), (w.grad, wr.grad, 'gw'), (b.grad, br.grad, 'gb')):
        assert float((a.float() - r).abs().max()) <= 4 * tol * float(r.abs().max()) + 1e-6, name


def test_hip_linear_module_trains():
    """A two-layer MLP of HipLinear modules, autograd + torch.optim.Adam: the loss goes down and the weights follow the same
    trajectory as the nn.Linear twin in f32."""
    O = _ops()
    torch.manual_seed(0)
    a = torch.nn.Sequential(O.HipLinear(64, 96, 'relu', compute_dtype=torch.float32), O.HipLinear(96, 8, 'none', compute_dtype=torch.float32)).to(DEV)
    b = torch.nn.Sequential(torch.nn.Linear(64, 96), torch.nn.ReLU(), torch.nn.Linear(96, 8)).to(DEV)
    b[0].load_state_dict(a[0].state_dict())
    b[2].load_state_dict(a[1].state_dict())
    oa, ob = torch.optim.Adam(a.parameters(), 1e-2), torch.optim.Adam(b.parameters(), 1e-2)
    x = torch.randn(512, 64, device=DEV)
    t = torch.randn(512, 8, device=DEV)
    first = None
    for i in range(20):
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            loss = ((net(x) - t) ** 2).mean()
            loss.backward()
            opt.step()
            if net is a and first is None:
                first = float(loss)
    assert float(loss) < first
    assert torch.allclose(a[0].weight, b[0].weight, rtol=1e-3, atol=1e-4) and torch.allclose(a[1].weight, b[2].weight, rtol=1e-3, atol=1e-4)


def test_rms_ops_follow_running_mean_std():
    O = _ops()
    g = torch.Generator().manual_seed(1)
    D = 253
    state = torch.zeros(2 * D + 1, dtype=torch.float64, device=DEV)
    state[D:] = 1.0
    mean, var, cnt = torch.zeros(D, dtype=torch.float64), torch.ones(D, dtype=torch.float64), 1.0
    for _ in range(3):
        x = (torch.randn(500, D, generator=g) * 2 + 1).to(DEV)
        y = torch.ops.ase_hip.rms_update_normalize(x, state)
        xc = x.cpu()
        bm, bv, n = xc.mean(0).double(), xc.var(0).double(), 500.0          # rl_games RunningMeanStd.forward (train)
        delta, tot = bm - mean, cnt + n
        mean, var, cnt = mean + delta * n / tot, (var * cnt + bv * n + delta ** 2 * cnt * n / tot) / tot, tot
        yr = torch.clamp((xc - mean.float()) / torch.sqrt(var.float() + 1e-5), -5, 5)
        assert torch.allclose(y.cpu(), yr, rtol=1e-4, atol=1e-5)
    # (the batch moments are f32 quantities in rl_games: torch's f32 mean and ours differ by summation order)
    assert torch.allclose(state[:D].cpu(), mean, rtol=1e-6, atol=1e-6) and float(state[2 * D]) == cnt
    x = torch.randn(64, D, generator=g).to(DEV)
    before = state.clone()
    y = torch.ops.ase_hip.rms_normalize(x, state)
    assert torch.equal(before, state)
    assert torch.allclose(y.cpu(), torch.clamp((x.cpu() - mean.float()) / torch.sqrt(var.float() + 1e-5), -5, 5), rtol=1e-4, atol=1e-5)


def test_gae_and_masked_norm():
    O = _ops()
    g = torch.Generator().manual_seed(2)
    H, N, gamma, tau = 16, 40, 0.99, 0.95
    dones = (torch.rand(H, N, generator=g) < 0.1).to(torch.uint8)
    values, nvalues, rewards = (torch.randn(H, N, 1, generator=g) for _ in range(3))
    advs, rets = torch.ops.ase_hip.gae(dones.to(DEV), values.to(DEV), nvalues.to(DEV), rewards.to(DEV), gamma, tau)
    last, ref = 0, torch.zeros(H, N, 1)
    for t in reversed(range(H)):                                          # learning/common_agent.py:437-449
        nd = (1.0 - dones[t].float()).unsqueeze(1)
        delta = rewards[t] + gamma * nvalues[t] - values[t]
        last = delta + gamma * tau * nd * last
        ref[t] = last
    assert torch.allclose(advs.cpu(), ref, rtol=1e-5, atol=1e-5) and torch.allclose(rets.cpu(), ref + values, rtol=1e-5, atol=1e-5)
    mask = (torch.rand(H * N, generator=g) < 0.7).float()
    adv = torch.ops.ase_hip.masked_norm(rets.view(-1), values.to(DEV).view(-1), mask.to(DEV))
    a = (rets.cpu() - values).view(-1)                                    # torch_ext.normalization_with_masks
    S = mask.sum()
    vm = a * mask
    m = vm.sum() / S
    msq = ((vm ** 2) / S).sum() - ((vm / S).sum()) ** 2
    std = torch.sqrt(msq * S / (S - 1))
    assert torch.allclose(adv.cpu(), (a - m) / (std + 1e-8), rtol=1e-4, atol=1e-5)


def test_small_ops():
    O = _ops()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(100, 64, generator=g).to(DEV)
    assert torch.allclose(torch.ops.ase_hip.normalize_rows(x), torch.nn.functional.normalize(x, dim=-1), rtol=1e-5, atol=1e-6)
    src = torch.randn(50, 31, generator=g).to(DEV)
    idx = torch.randint(0, 50, (200,), generator=g).to(torch.int32).to(DEV)
    assert torch.equal(torch.ops.ase_hip.gather_rows(src, idx), src[idx.long()])
    lg = torch.randn(77, 1, generator=g).to(DEV)
    p = 1 / (1 + torch.exp(-lg))
    ref = -torch.log(torch.maximum(1 - p, torch.tensor(0.0001, device=DEV))) * 2.0           # learning/amp_agent.py:563-570
    assert torch.allclose(torch.ops.ase_hip.disc_reward(lg, 2.0), ref, rtol=1e-4, atol=1e-5)
    st = torch.tensor([7, 0], dtype=torch.int64, device=DEV)
    z = torch.ops.ase_hip.sample_latents(1000, 64, st)
    assert int(st[1]) == 1 and torch.allclose(z.norm(dim=-1), torch.ones(1000, device=DEV), atol=1e-5)
    with pytest.raises(RuntimeError):
        torch.ops.ase_hip.linear_act(torch.zeros(4, 8, device=DEV, dtype=torch.float64), torch.zeros(3, 8, device=DEV), torch.zeros(3, device=DEV), 'relu')


def _neglogp(a, mu, logstd):
    return 0.5 * (((a - mu) / torch.exp(logstd)) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * a.shape[-1] + logstd.sum(-1)


@pytest.mark.parametrize('masked,clip_value', [(True, False), (False, True)])
def test_ppo_loss_head_op_against_autograd(masked, clip_value):
    """ase_hip::ppo_loss_head against the reference's formulas written in plain PyTorch (learning/common_agent.py:456-464,505-534,
    learning/amp_agent.py:316-324; rl_games neglogp / policy_kl) and autograd's gradients of them."""
    _ops()
    g = torch.Generator().manual_seed(5)
    M, A, e_clip, cc, bc = 1000, 31, 0.2, 5.0, 10.0
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    mu = (r(M, A) * 0.8).requires_grad_()
    value = r(M, 1).requires_grad_()
    old_mu = mu.detach() + 0.05 * r(M, A)
    logstd = torch.full((A,), -2.9, device=DEV)
    sigma = torch.exp(logstd).expand(M, A).contiguous()
    actions = old_mu + sigma * r(M, A)
    old_nlp = _neglogp(actions, old_mu, logstd)
    adv, old_v, ret = r(M), value.detach().view(-1) + 0.3 * r(M), r(M)
    mask = (torch.rand(M, generator=g) < 0.7).float().to(DEV) if masked else torch.empty(0, device=DEV)
    stats, d_mu, d_v = torch.ops.ase_hip.ppo_loss_head(mu.detach(), value.detach(), actions, old_mu, sigma, old_nlp, adv, old_v, ret,
                                                       mask, logstd, e_clip, cc, bc, clip_value)
    w = mask if masked else torch.ones(M, device=DEV)
    mean = lambda x: (x * w).sum() / w.sum()
    ratio = torch.exp(old_nlp - _neglogp(actions, mu, logstd))
    a_loss = mean(torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1 - e_clip, 1 + e_clip)))
    b_loss = mean((torch.clamp_min(mu - 1.0, 0) ** 2 + torch.clamp_max(mu + 1.0, 0) ** 2).sum(-1))
    v = value.view(-1)
    if clip_value:
        vpc = old_v + (v - old_v).clamp(-e_clip, e_clip)
        c_loss = torch.max((v - ret) ** 2, (vpc - ret) ** 2).mean()
    else:
        c_loss = ((ret - v) ** 2).mean()
    ent = mean((0.5 + 0.5 * math.log(2 * math.pi) + logstd).sum().expand(M))
    kl = (torch.log(sigma / sigma + 1e-5) + (sigma ** 2 + (old_mu - mu.detach()) ** 2) / (2 * (sigma ** 2 + 1e-5)) - 0.5).sum(-1).mean()
    clipf = mean(((ratio.detach() - 1).abs() > e_clip).float())
    ref = torch.stack([a_loss, c_loss, b_loss, ent, clipf, kl]).detach()
    assert torch.allclose(stats, ref, rtol=2e-5, atol=1e-6), (stats, ref)
    gm, gv = torch.autograd.grad(a_loss + bc * b_loss + cc * c_loss, [mu, value])
    assert torch.allclose(d_mu, gm, rtol=1e-4, atol=1e-7 * float(gm.abs().max()) + 1e-10)
    assert torch.allclose(d_v, gv, rtol=1e-4, atol=1e-9)


def test_disc_and_enc_div_loss_ops_against_autograd():
    """ase_hip::disc_loss_gp and ase_hip::enc_div_loss against learning/amp_agent.py:442-459,481-496 and
    learning/ase_agent.py:413-418,445-467 in plain PyTorch."""
    _ops()
    g = torch.Generator().manual_seed(9)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)
    n, D, dc = 512, 140, 5.0
    logits = r(3 * n, 1).requires_grad_()
    grad_demo = 0.1 * r(n, D)
    stats, d_l = torch.ops.ase_hip.disc_loss_gp(logits.detach(), grad_demo, dc)
    bce = torch.nn.functional.binary_cross_entropy_with_logits
    la, ld = logits[:2 * n], logits[2 * n:]
    loss = 0.5 * (bce(la, torch.zeros_like(la)) + bce(ld, torch.ones_like(ld)))
    ref = torch.stack([loss.detach(), (grad_demo ** 2).sum(-1).mean(), (la < 0).float().mean(), (ld > 0).float().mean()])
    assert torch.allclose(stats, ref, rtol=2e-5, atol=1e-6), (stats, ref)
    assert torch.allclose(d_l, torch.autograd.grad(dc * loss, logits)[0], rtol=1e-4, atol=1e-9)
    stats0, _ = torch.ops.ase_hip.disc_loss_gp(logits.detach(), torch.empty(0, device=DEV), dc)
    assert float(stats0[1]) == 0.0
    # encoder + diversity
    M, A, Z, ec, dvc, tar = 768, 31, 64, 5.0, 0.01, 1.0
    e = r(n, Z).requires_grad_()
    ez = torch.nn.functional.normalize(r(n, Z), dim=-1)
    mu, mu2 = (1.2 * r(M, A)).requires_grad_(), (1.2 * r(M, A)).requires_grad_()
    z, zn = torch.nn.functional.normalize(r(M, Z), dim=-1), torch.nn.functional.normalize(r(M, Z), dim=-1)
    st, d_e, d_mu, d_mu2 = torch.ops.ase_hip.enc_div_loss(e.detach(), ez, mu.detach(), mu2.detach(), z, zn, ec, dvc, tar)
    enc_loss = (-(torch.nn.functional.normalize(e, dim=-1) * ez).sum(-1)).mean()
    a_diff = ((mu.clamp(-1, 1) - mu2.clamp(-1, 1)) ** 2).mean(-1)
    z_diff = 0.5 - 0.5 * (zn * z).sum(-1)
    div_loss = ((tar - a_diff / (z_diff + 1e-5)) ** 2).mean()
    assert torch.allclose(st, torch.stack([enc_loss, div_loss]).detach(), rtol=2e-5, atol=1e-6), st
    assert torch.allclose(d_e, torch.autograd.grad(ec * enc_loss, e)[0], rtol=1e-4, atol=1e-9)
    gm, gm2 = torch.autograd.grad(dvc * div_loss, [mu, mu2])
    assert torch.allclose(d_mu, gm, rtol=1e-4, atol=1e-9) and torch.allclose(d_mu2, gm2, rtol=1e-4, atol=1e-9)
