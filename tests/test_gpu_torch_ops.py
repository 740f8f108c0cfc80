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
