"""Shared helpers for the engine tests (CPU emulator and GPU)."""
import torch

from ase_amd.learning.network_builder import AMPBuilder, ASEBuilder, HRLBuilder

BUILDERS = {'ase': ASEBuilder, 'amp': AMPBuilder, 'ppo': HRLBuilder}


def build_net(G, device='cpu'):
    spec, kind = G['spec'], G['kind']
    b = BUILDERS[kind]()
    b.load(G['net'])
    kw = dict(actions_num=spec['act_size'], input_shape=(spec['obs_size'],), num_seqs=spec['num_envs'], value_size=1,
              device=device)
    if kind in ('amp', 'ase'):
        kw['amp_input_shape'] = (spec['amp_obs_size'],)
    if kind == 'ase':
        kw['ase_latent_shape'] = (G['cfg']['latent_dim'],)
    net = b.build(kind, **kw)
    net.load_state_dict({k: v.to(device) for k, v in golden_init_sd(G).items()})
    return net


def set_rms(state_vec, s):
    D = s['mean'].numel()
    state_vec[:D] = s['mean'].to(state_vec.device)
    state_vec[D:2 * D] = s['var'].to(state_vec.device)
    state_vec[2 * D] = s['count'].to(state_vec.device)


def get_rms(state_vec):
    D = (state_vec.numel() - 1) // 2
    return {'mean': state_vec[:D].cpu(), 'var': state_vec[D:2 * D].cpu(), 'count': state_vec[2 * D].cpu()}


def close(a, b, rtol, atol, what=''):
    a, b = torch.as_tensor(a).detach().cpu().double(), torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    ok = err <= atol + rtol * b.abs()
    assert bool(ok.all()), (what, 'max abs err', float(err.max()), 'ref max', float(b.abs().max()),
                            'bad', int((~ok).sum()), 'of', ok.numel())


# ---- compact goldens at the real layer widths (oracle/make_golden.py: seeded_init / sample) ------------------------------
def seeded_init(shapes, bounds, seed):
    """Initial weights of a real-width golden: every weight matrix U(-b, b) with b = the bound the REFERENCE's own
    initialiser used for that tensor (recorded by make_golden.py as max |w| rounded up), biases / sigma as the reference
    left them (zeros / the configured constant).  A function of (shapes, bounds, seed) only, so 7 M weights cost nothing
    in the fixture: make_golden.py loads the same tensors into the reference agent before it runs."""
    g = torch.Generator().manual_seed(int(seed) * 7919 + 17)
    sd = {}
    for k, shp in shapes.items():
        b = bounds[k]
        if isinstance(b, dict):                       # constant tensor (biases: 0, sigma: its initial value)
            sd[k] = torch.full(tuple(shp), float(b['const']))
        else:
            sd[k] = (torch.rand(tuple(shp), generator=g) * 2.0 - 1.0) * float(b)
    return sd


def sample_index(name, numel, n, seed):
    import zlib
    g = torch.Generator().manual_seed((int(seed) * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
    return torch.randint(0, numel, (min(int(n), numel),), generator=g)


def pack_sampled(name, t, n, seed):
    """What a compact golden keeps of a large tensor: its L2 norm (f64), its sum (f64) and n seeded elements."""
    t = t.detach().reshape(-1)
    return {'sampled': True, 'norm': float(t.double().norm()), 'sum': float(t.double().sum()),
            'vals': t[sample_index(name, t.numel(), n, seed)].clone(), 'numel': t.numel()}


def close_entry(name, actual, entry, rtol, atol, seed, what=''):
    """Compare a tensor with a golden entry: a full tensor, or a pack_sampled() record (sample element-wise at (rtol, atol),
    norm at 4 x rtol)."""
    if isinstance(entry, dict) and entry.get('sampled'):
        a = torch.as_tensor(actual).detach().cpu().reshape(-1)
        assert a.numel() == entry['numel'], (what, a.numel(), entry['numel'])
        close(a[sample_index(name, a.numel(), entry['vals'].numel(), seed)], entry['vals'], rtol, atol, what + ' (sample)')
        nrm = float(a.double().norm())
        assert abs(nrm - entry['norm']) <= 4 * rtol * entry['norm'] + atol * a.numel() ** 0.5, (what, 'norm', nrm, entry['norm'])
    else:
        close(actual, entry, rtol, atol, what)


def golden_init_sd(G):
    """Initial state_dict of a golden: stored, or regenerated from its seed."""
    if 'init_sd' in G:
        return G['init_sd']
    return seeded_init(G['init_shapes'], G['init_bounds'], G['init_seed'])
