"""GPU parity tests, op level: every C-ABI entry point of libase_hip.so against the CPU emulation of the
same op (tests/emu_backend.py) on identical seeded inputs.  Shapes include the odd sizes of the real
nets (K = 253+64 -> 320 padded concat, 1400 -> 1408, N = 31 / 1 heads, ragged M).
f32 storage = exact-f32 MFMA: tolerance is summation-order noise; bf16 / f16 storage: inputs are rounded to
the storage type on both sides, products accumulate in f32, so only the output rounding (2^-8 relative) remains."""
import math

import pytest
import torch

from ase_amd import lib as L
from tests.emu_backend import EmuBackend
from tests.helpers import close

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16, torch.float16]


@pytest.fixture(scope='module')
def be():
    from ase_amd.backend import HipBackend
    return HipBackend()


def _pair(t):
    return t.cuda(), t.clone()


def _tol(dt):
    return (2e-5, 2e-5) if dt == torch.float32 else ((1.2e-2, 2e-3) if dt == torch.bfloat16 else (1.5e-3, 3e-4))


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 64, 320), (77, 32, 128), (513, 320, 1408), (1024, 1024, 1024),
                                   (4096, 512, 1024), (33, 1408, 1024)])
def test_gemm_nt_plain(be, dt, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dt)
    B = (torch.randn(N, K, generator=g) * 0.1).to(dt)
    bias = torch.randn(N, generator=g)
    Ag, Ac = _pair(A)
    Bg, Bc = _pair(B)
    bg, bc = _pair(bias)
    Cg, Cc = torch.zeros(M, N, dtype=dt).cuda(), torch.zeros(M, N, dtype=dt)
    be.gemm_nt(Ag, Bg, Cg, M, N, K, bias=bg, act=L.ACT_RELU)
    EmuBackend().gemm_nt(Ac, Bc, Cc, M, N, K, bias=bc, act=L.ACT_RELU)
    rt, at = _tol(dt)
    close(Cg.float(), Cc.float(), rt, at * math.sqrt(K / 64), f'nt {M}x{N}x{K}')


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('act,aux_mode', [(L.ACT_TANH, L.AUX_NONE), (L.ACT_NONE, L.AUX_RELU_MASK),
                                          (L.ACT_NONE, L.AUX_TANH_GRAD)])
def test_gemm_nt_epilogues(be, dt, act, aux_mode):
    M, N, K = 300, 192, 256
    g = torch.Generator().manual_seed(11)
    A = (torch.randn(M, K + 64, generator=g) * 0.3).to(dt)[:, :K]          # lda > K
    B = (torch.randn(N, K, generator=g) * 0.1).to(dt)
    aux = (torch.randn(M, N, generator=g)).clamp(-0.9, 0.9).to(dt)
    Cbig = torch.zeros(M, N + 64, dtype=dt)
    cs = torch.zeros(N + 8)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        Ad, Bd, auxd = A.to(dev), B.to(dev), aux.to(dev)
        Cd, csd = Cbig.to(dev).clone(), cs.to(dev).clone()
        Cv = Cd[:, 64:]                                                      # column-offset output view
        b.gemm_nt(Ad, Bd, Cv, M, N, K, aux=auxd if aux_mode else None, aux_mode=aux_mode, colsum=csd, colsum_n=N - 5,
                  act=act, alpha=0.5)
        outs.append((Cd.float().cpu(), csd.cpu()))
    rt, at = _tol(dt)
    close(outs[0][0], outs[1][0], rt, at * 2, 'C')
    close(outs[0][1], outs[1][1], rt * 5, at * 40, 'colsum')
    assert float(outs[0][1][N - 5:].abs().max()) == 0.0
    assert float(outs[0][0][:, :64].abs().max()) == 0.0


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('act', [L.ACT_SILU, L.ACT_ELU, L.ACT_GELU, L.ACT_SIGMOID, L.ACT_SELU, L.ACT_SOFTPLUS])
def test_gemm_nt_smooth_activations(be, dt, act):
    """Row X1: the rl_games activations beyond relu / tanh (swish = SiLU, elu, gelu, sigmoid, selu, softplus): the forward
    epilogue stores act(z) and keeps the pre-activation z as the layer's twin; the data-gradient epilogue multiplies by
    act'(z) read back from that twin (ASE_AUX_PREACT); the gradient-penalty kernels use act''(z) / act'(z)^2."""
    M, N, K = 300, 192, 256
    g = torch.Generator().manual_seed(100 + act)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dt)
    B = (torch.randn(N, K, generator=g) * 0.15).to(dt)
    bias = torch.randn(N, generator=g) * 0.5
    dY = (torch.randn(M, N, generator=g) * 0.3).to(dt)
    W2 = (torch.randn(N, N, generator=g) * 0.1).to(dt)
    wl = torch.randn(N, generator=g)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        H, Z = torch.zeros(M, N, dtype=dt, device=dev), torch.zeros(M, N, dtype=dt, device=dev)
        b.gemm_nt(A.to(dev), B.to(dev), H, M, N, K, bias=bias.to(dev), act=act, mask_out=Z)
        dX = torch.zeros(M, N, dtype=dt, device=dev)
        b.gemm_nt(dY.to(dev), W2.to(dev), dX, M, N, N, aux=Z, aux_mode=L.AUX_PREACT | (act << 8))
        gs = torch.zeros(M, N, dtype=dt, device=dev)
        b.gp_seed(Z, wl.to(dev), gs, M, N, scale=0.7, act=act)
        dz = dY.to(dev).clone()
        b.gp_second(Z, gs, dX, dz, M, N, act)
        outs.append([t.float().cpu() for t in (H, Z, dX, gs, dz)])
    rt, at = _tol(dt)
    for name, a, c in zip(('act(z)', 'z twin', 'dX = (dY W) act\'(z)', 'gp seed', 'gp second'), *outs):
        close(a, c, rt * 2, at * 4, name)


@pytest.mark.parametrize('dt', DT)
def test_gemm_nt_f32_out(be, dt):
    M, N, K = 257, 64, 512
    g = torch.Generator().manual_seed(5)
    A = (torch.randn(M, K, generator=g) * 0.3).to(dt)
    B = (torch.randn(N, K, generator=g) * 0.1).to(dt)
    Cg, Cc = torch.zeros(M, N).cuda(), torch.zeros(M, N)
    be.gemm_nt(A.cuda(), B.cuda(), Cg, M, N, K)
    EmuBackend().gemm_nt(A, B, Cc, M, N, K)
    close(Cg, Cc, 2e-5, 2e-5, 'f32 out')


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('M,N,K,nr,kr,ss,sd', [(256, 128, 128, 128, 128, 128, 128), (1000, 64, 320, 48, 317, 253, 256),
                                               (4096, 1024, 1408, 1024, 1400, 1400, 1400), (130, 64, 64, 1, 40, 40, 40),
                                               (16384, 512, 1024, 512, 1024, 1024, 1024), (50, 128, 64, 100, 53, 37, 40)])
def test_gemm_tn(be, dt, M, N, K, nr, kr, ss, sd):
    g = torch.Generator().manual_seed(M + N + K)
    A = (torch.randn(M, N, generator=g) * 0.2).to(dt)
    B = (torch.randn(M, K, generator=g) * 0.2).to(dt)
    G0 = torch.randn(nr, kr, generator=g)
    b0 = torch.randn(nr, generator=g)
    Gg, Gc = G0.cuda(), G0.clone()
    bg, bc = b0.cuda(), b0.clone()
    be.gemm_tn(A.cuda(), B.cuda(), Gg, M, N, K, nr, kr, ss, sd, alpha=0.7, gbias=bg)
    EmuBackend().gemm_tn(A, B, Gc, M, N, K, nr, kr, ss, sd, alpha=0.7, gbias=bc)
    close(Gg, Gc, 3e-5, 3e-5 * math.sqrt(M), f'tn {M}x{N}x{K}')
    close(bg, bc, 3e-5, 3e-5 * math.sqrt(M), f'tn bias {M}x{N}')


@pytest.mark.parametrize('dt', DT)
def test_refresh_shadow(be, dt):
    n, k, ss, sd = 48, 53, 37, 64
    W = torch.randn(n, k)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        Ws, Wts = torch.zeros(64, 128, dtype=dt, device=dev), torch.zeros(128, 64, dtype=dt, device=dev)
        b.refresh_shadow(W.to(dev), Ws, Wts, ss, sd)
        outs.append((Ws.float().cpu(), Wts.float().cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(outs[0][0].t(), outs[0][1])


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('D,W,gathered', [(253, 320, True), (1400, 1408, True), (1400, 1408, False)])   # scalar / 16-byte paths
def test_rms_pipeline(be, dt, D, W, gathered):
    H, N, M = 8, 50, 300
    g = torch.Generator().manual_seed(3)
    src = torch.randn(H * N, D, generator=g) * 2 + 0.5
    idx = torch.randperm(H * N, generator=g)[:M].to(torch.int32) if gathered else None
    remap = (H, N) if gathered else (0, 0)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        state = torch.zeros(2 * D + 1, dtype=torch.float64, device=dev)
        state[D:] = 1.0
        state[:D] = 0.1
        sums = torch.zeros(3, 2 * D, dtype=torch.float64, device=dev)
        mean, std = torch.zeros(3, D, device=dev), torch.zeros(3, D, device=dev)
        o0 = torch.zeros(M, W, dtype=dt, device=dev)
        o1 = torch.zeros(M, W, dtype=dt, device=dev)
        for s in range(3):
            b.rms_moments(src.to(dev), D, None if idx is None else idx.to(dev), remap, M, state, sums[s])
        b.rms_finalize(state, D, sums, M, 3, mean, std)
        b.rms_normalize(src.to(dev), D, None if idx is None else idx.to(dev), remap, M, mean[2], std[2], [o0, o1[:, 0:]])
        outs.append((state.cpu(), mean.cpu(), std.cpu(), o0.float().cpu(), o1.float().cpu()))
    close(outs[0][0], outs[1][0], 1e-9, 1e-12, 'state')
    close(outs[0][1], outs[1][1], 1e-6, 1e-7, 'mean')
    close(outs[0][2], outs[1][2], 1e-6, 1e-7, 'std')
    tol = 1e-5 if dt == torch.float32 else 1e-2
    close(outs[0][3], outs[1][3], tol, tol, 'normalised')
    assert torch.equal(outs[0][3], outs[0][4])


def test_rms_eval_and_unnorm(be):
    D = 17
    st = torch.rand(2 * D + 1, dtype=torch.float64, generator=torch.Generator().manual_seed(3)) + 0.5
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        mean, std = torch.zeros(1, D, device=dev), torch.zeros(1, D, device=dev)
        b.rms_finalize(st.to(dev), D, None, 0, 0, mean, std)
        x = torch.linspace(-7, 7, 1000).to(dev)
        y = torch.zeros_like(x)
        b.rms_unnormalize(st[[0, D, 2 * D]].contiguous().to(dev), x, y)
        outs.append((mean.cpu(), std.cpu(), y.cpu()))
    for a, c in zip(*outs):
        close(a, c, 3e-7, 1e-7)        # (sqrtf on the device is within 1 ulp, not correctly rounded)


def test_colsum(be):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4096, 576, generator=g)
    out0 = torch.randn(512, generator=g)
    og, oc = out0.clone().cuda(), out0.clone()
    be.colsum(x.cuda()[:, :512], 4096, 500, og, scale=0.25)          # row pitch 576, 500 of 512 columns
    EmuBackend().colsum(x[:, :512], 4096, 500, oc, scale=0.25)
    close(og, oc, 1e-5, 1e-4, 'colsum')
    assert torch.equal(og[500:].cpu(), out0[500:])


def test_gather_rows(be):
    src = torch.randn(640, 31)
    idx = torch.randperm(640)[:200].to(torch.int32)
    for dt in DT:
        dg, dc = torch.zeros(200, 64, dtype=dt).cuda(), torch.zeros(200, 64, dtype=dt)
        be.gather_rows(src.cuda(), 31, idx.cuda(), (16, 40), 200, dg)
        EmuBackend().gather_rows(src, 31, idx, (16, 40), 200, dc)
        assert torch.equal(dg.cpu(), dc)


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16, torch.float32])
def test_gather_multi_identity_map_is_a_conversion(be, dt):
    """ase_hip_gather_multi with a NULL row map: several f32 matrices -> storage type in one launch (the gradient-penalty
    chain handed to the 16-bit launches) - 8 values per thread where the field allows it, element-wise otherwise (odd width,
    a view that starts off a 16-byte boundary); saturating for half."""
    g = torch.Generator().manual_seed(3)
    M = 777
    srcs = [torch.randn(M, 1024, generator=g) * 3, torch.randn(M, 1408, generator=g) * 1e-3, torch.randn(M, 31, generator=g),
            torch.randn(M, 520, generator=g)[:, 4:516]]
    srcs[0][5, 7] = 1e6                                     # beyond half's range
    srcs = [t.cuda() for t in srcs]
    dsts = [torch.zeros(M, 1024, dtype=dt).cuda(), torch.zeros(M, 1472, dtype=dt).cuda()[:, :1408], torch.zeros(M, 64, dtype=dt).cuda(),
            torch.zeros(M, 512, dtype=dt).cuda()]
    code = {torch.bfloat16: L.BF16, torch.float16: L.F16, torch.float32: L.F32}[dt]
    items = [(s_, s_.shape[1], d_) for s_, d_ in zip(srcs, dsts)]
    desc = torch.tensor([[s_.data_ptr(), s_.stride(0), c, d_.data_ptr(), d_.stride(0), code] for s_, c, d_ in items],
                        dtype=torch.int64, device='cuda')
    be.gather_multi(desc, items, None, (0, 0), M)
    torch.cuda.synchronize()
    for s_, c, d_ in items:
        ref = s_[:, :c].clamp(-65504, 65504).to(dt) if dt == torch.float16 else s_[:, :c].to(dt)
        assert torch.equal(d_[:, :c], ref), (dt, c)
    assert float(dsts[2][:, 31:].abs().max()) == 0.0        # columns past the field's width are left alone


def _mb(M, D, Z, g, masked=True):
    mb = {'actions': torch.randn(M, D, generator=g) * 0.3, 'mu': torch.randn(M, D, generator=g) * 0.3,
          'sigma': torch.full((M, D), math.exp(-2.9)), 'old_logp_actions': torch.randn(M, 1, generator=g) * 2 - 60,
          'advantages': torch.randn(M, 1, generator=g), 'old_values': torch.randn(M, 1, generator=g),
          'returns': torch.randn(M, 1, generator=g)}
    if masked:
        mb['rand_action_mask'] = (torch.rand(M, 1, generator=g) < 0.8).float()
    if Z:
        z = torch.randn(M, Z, generator=g)
        mb['ase_latents'] = z / z.norm(dim=-1, keepdim=True)
    return mb


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('masked,div_on,mu_tanh,clip_value', [(1, 1, 0, 0), (1, 0, 0, 1), (0, 0, 1, 0)])
@pytest.mark.parametrize('D', [31, 64])        # humanoid actions / the HRL high-level action (= the 64-d latent)
def test_ppo_head(be, dt, masked, div_on, mu_tanh, clip_value, D):
    M, Z = 1003, 64
    g = torch.Generator().manual_seed(17 + masked + 2 * div_on)
    mb = _mb(M, D, Z if div_on else 0, g, masked)
    rows = 2 * M if div_on else M
    mu = torch.zeros(rows, 64)
    mu[:, :D] = torch.randn(rows, D, generator=g) * 0.7
    # keep the importance ratio in a range where all three surrogate branches occur
    logstd = torch.full((D,), -2.9)
    with torch.no_grad():
        m = torch.tanh(mu[:M, :D]) if mu_tanh else mu[:M, :D]
        # realistic magnitudes: a = m + sigma * eps, so that neglogp is O(10) and f32 rounding of it is ~1e-6
        mb['actions'] = m + math.exp(-2.9) * torch.randn(M, D, generator=g) * 1.5
        nlp = 0.5 * (((mb['actions'] - m) / math.exp(-2.9)) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * D + logstd.sum()
        mb['old_logp_actions'] = (nlp + torch.randn(M, generator=g) * 0.3).view(M, 1)
    value = torch.zeros(M, 64)
    value[:, 0] = torch.randn(M, generator=g)
    new_z = None
    if div_on:
        nz = torch.randn(M, Z, generator=g)
        new_z = nz / nz.norm(dim=-1, keepdim=True)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        d_mu, d_v = torch.zeros(rows, 64, dtype=dt, device=dev), torch.zeros(M, 64, dtype=dt, device=dev)
        dbm, dbv = torch.zeros(D, device=dev), torch.zeros(1, device=dev)
        acc = torch.zeros(L.ACC_COUNT, dtype=torch.float64, device=dev)
        if masked:
            acc[L.ACC_MASK_SUM] = float(mb['rand_action_mask'].sum())
        mbd = {k: v.to(dev) for k, v in mb.items()}
        b.ppo_head(mu.to(dev), value.to(dev), mbd, None if new_z is None else new_z.to(dev), logstd.to(dev), d_mu, d_v,
                   dbm, dbv, acc, M, M, D, Z if div_on else 0, masked, div_on, mu_tanh, clip_value, 0.2, 5.0, 10.0, 0.01,
                   1.0)
        outs.append((d_mu.float().cpu(), d_v.float().cpu(), dbm.cpu(), dbv.cpu(), acc.cpu()))
    t = 2e-5 if dt == torch.float32 else 1e-2
    sc = float(outs[1][0].abs().max())
    close(outs[0][0], outs[1][0], t * 5, t * sc, 'd_mu')
    close(outs[0][1], outs[1][1], t * 5, t * float(outs[1][1].abs().max()), 'd_value')
    close(outs[0][2], outs[1][2], t * 50, t * 50 * sc, 'db_mu')
    close(outs[0][3], outs[1][3], t * 50, t * 5 * float(outs[1][1].abs().max()), 'db_value')
    # sums of signed surrogate terms: expf/logf differ by an ulp between host and device and the sum cancels
    a0, a1 = outs[0][4].clone(), outs[1][4].clone()
    assert abs(float(a0[L.ACC_CLIPPED] - a1[L.ACC_CLIPPED])) <= 3      # a count: samples exactly at the clip edge
    a0[L.ACC_CLIPPED] = a1[L.ACC_CLIPPED] = 0
    close(a0, a1, 5e-4, 2e-2, 'acc')


@pytest.mark.parametrize('dt', DT)
def test_disc_enc_heads(be, dt):
    amb, Z = 333, 64
    g = torch.Generator().manual_seed(23)
    HD = torch.zeros(3 * amb, 128)
    HD[:, 0] = torch.randn(3 * amb, generator=g) * 3
    HD[:amb, 64:] = torch.randn(amb, Z, generator=g)
    z = torch.randn(amb, Z, generator=g)
    z = z / z.norm(dim=-1, keepdim=True)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        dHD = torch.zeros(3 * amb, 128, dtype=dt, device=dev)
        dbl, dbe = torch.zeros(1, device=dev), torch.zeros(Z, device=dev)
        enc_out = torch.zeros(amb, Z, device=dev)
        acc = torch.zeros(L.ACC_COUNT, dtype=torch.float64, device=dev)
        hd = HD.to(dev)
        b.disc_head(hd, dHD, dbl, acc, amb, amb, 5.0)
        b.enc_head(hd[:amb, 64:], z.to(dev), dHD[:amb, 64:], dbe, enc_out, acc, amb, amb, Z, 5.0)
        outs.append((dHD.float().cpu(), dbl.cpu(), dbe.cpu(), enc_out.cpu(), acc.cpu()))
    t = 2e-5 if dt == torch.float32 else 1e-2
    sc = float(outs[1][0].abs().max())
    close(outs[0][0], outs[1][0], t * 5, t * sc, 'dHD')
    close(outs[0][1], outs[1][1], t * 50, t * 20 * sc, 'db_logit')
    close(outs[0][2], outs[1][2], t * 50, t * 20 * sc, 'db_enc')
    close(outs[0][3], outs[1][3], 1e-5, 1e-6, 'enc_out')
    close(outs[0][4], outs[1][4], 2e-5, 1e-6, 'acc')


@pytest.mark.parametrize('dt', DT)
def test_enc_grad_penalty_row_ops(be, dt):
    """ase_hip_enc_gp_seed / ase_hip_enc_gp_back (learning/ase_agent.py:431-441) against the emulator's
    formulas on strided views of a joint head buffer, and the Jacobian against autograd of u(e) in f64."""
    amb, Z, off = 333, 64, 64
    g = torch.Generator().manual_seed(29)
    HD = torch.zeros(amb, 128)
    HD[:, off:] = torch.randn(amb, Z, generator=g) * 2
    z = torch.randn(amb, Z, generator=g)
    z = z / z.norm(dim=-1, keepdim=True)
    DU = torch.zeros(amb, 128)
    DU[:, off:] = torch.randn(amb, Z, generator=g)
    d0 = (torch.randn(amb, 128, generator=g) * 0.1).to(dt)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        U = torch.zeros(amb, 128, dtype=dt, device=dev)
        dHD = d0.clone().to(dev)
        dbe = torch.zeros(Z, device=dev)
        hd, zz, du = HD.to(dev), z.to(dev), DU.to(dev)
        b.enc_gp_seed(hd[:, off:], zz, U[:, off:], amb, Z, scale=0.7)
        b.enc_gp_back(hd[:, off:], zz, du[:, off:], dHD[:, off:], dbe, amb, Z)
        outs.append((U.float().cpu(), dHD.float().cpu(), dbe.cpu()))
    t = 2e-5 if dt == torch.float32 else 1e-2
    assert float(outs[0][0][:, :off].abs().max()) == 0.0                 # only the encoder columns are written
    close(outs[0][0], outs[1][0], t, t * float(outs[1][0].abs().max()), 'u')
    close(outs[0][1], outs[1][1], t, t * float(outs[1][1].abs().max()), 'd_e')
    close(outs[0][2], outs[1][2], t * 50, t * 20 * float(outs[1][1].abs().max()), 'db_enc')
    assert torch.equal(outs[0][1][:, :off], d0.float()[:, :off])
    if dt == torch.float32:
        e = HD[:, off:].double().requires_grad_(True)
        zz = z.double()
        n = e.norm(dim=-1, keepdim=True)
        u = -(zz - e / n * ((e / n) * zz).sum(-1, keepdim=True)) / n
        jr = torch.autograd.grad(u, e, grad_outputs=DU[:, off:].double())[0]      # J is symmetric: J du = (du^T J)^T
        close(outs[0][1][:, off:].double() - d0.float()[:, off:].double(), jr, 1e-4, 1e-5 * float(jr.abs().max()), 'J du vs autograd')


@pytest.mark.parametrize('dt', DT)
def test_gp_seed_sqnorm_reduce(be, dt):
    rows, width = 200, 512
    h = torch.randn(rows, 576).to(dt)
    w = torch.randn(width)
    x = torch.randn(300, 1408).to(dt)
    v = torch.randn(100000)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        g = torch.zeros(rows, 512, dtype=dt, device=dev)
        acc = torch.zeros(L.ACC_COUNT, dtype=torch.float64, device=dev)
        b.gp_seed(h.to(dev), w.to(dev), g, rows, width)
        b.sqnorm(x.to(dev), 300, 1400, acc, L.ACC_GP)
        b.reduce_sum(v.to(dev), v.numel(), False, acc, 0)
        b.reduce_sum(v.to(dev), v.numel(), True, acc, 1)
        outs.append((g.float().cpu(), acc.cpu()))
    assert torch.equal(outs[0][0], outs[1][0])
    a_g, a_c = outs[0][1].clone(), outs[1][1].clone()
    close(a_g[L.ACC_GP], a_c[L.ACC_GP], 1e-6, 0, 'sqnorm')      # 16-byte chunks are summed in f32 before the f64 accumulation
    a_g[L.ACC_GP] = a_c[L.ACC_GP] = 0
    close(a_g, a_c, 1e-10, 1e-9, 'acc')


def test_finalize_begin_adam_axpy(be):
    cfg = dict(critic_coef=5, entropy_coef=0.0, bounds_loss_coef=10, disc_coef=5, disc_logit_reg=0.01,
               disc_grad_penalty=5, disc_weight_decay=1e-4, enc_coef=5, enc_weight_decay=0.0, amp_diversity_bonus=0.01,
               enc_grad_penalty=3.0)
    n = 100003
    g = torch.Generator().manual_seed(2)
    w0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 1e-3
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        acc = (torch.arange(L.ACC_COUNT, dtype=torch.float64) + 1.5).to(dev)
        res = torch.zeros(L.RES_COUNT, device=dev)
        b.finalize_scalars(acc, res, 1000, 250, 1, 1, 1, 1, cfg)
        st = torch.tensor([0.0, 2e-5, 0.9, 0.999, 1e-8, 1.0, 1.0, 0.0], dtype=torch.float64, device=dev)
        w, m, v, gd = w0.to(dev).clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), gr.to(dev).clone()
        for _ in range(3):
            b.begin_step(st, acc)
            b.axpy(gd, w, 1e-3)
            b.adam(w, gd, m, v, st)
        outs.append((res.cpu(), st.cpu(), acc.cpu(), w.cpu(), m.cpu(), v.cpu()))
    close(outs[0][0], outs[1][0], 1e-6, 1e-7, 'res')
    close(outs[0][1], outs[1][1], 1e-12, 1e-15, 'opt_state')
    assert float(outs[0][2].abs().max()) == 0.0
    close(outs[0][3], outs[1][3], 1e-6, 1e-9, 'w')
    close(outs[0][4], outs[1][4], 1e-6, 1e-12, 'm')
    close(outs[0][5], outs[1][5], 1e-6, 1e-15, 'v')


def test_rollout_tail_ops(be):
    H, N, Z, A = 32, 300, 64, 140
    g = torch.Generator().manual_seed(8)
    HD = torch.randn(H * N, 128, generator=g) * 3
    z = torch.randn(H * N, Z, generator=g)
    z = z / z.norm(dim=-1, keepdim=True)
    dones = (torch.rand(H, N, generator=g) < 0.05).to(torch.uint8)
    val, nval, rt = torch.randn(H, N, 1, generator=g), torch.randn(H, N, 1, generator=g), torch.ones(H, N, 1)
    mask = (torch.rand(H, N, generator=g) < 0.7).float()
    amp = torch.randn(H * N, A, generator=g)
    idx = torch.randperm(H * N, generator=g)[:500].to(torch.int32)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        rd, re = torch.zeros(H * N, device=dev), torch.zeros(H * N, device=dev)
        hd = HD.to(dev)
        b.disc_reward(hd, rd, H * N, 2.0)
        b.enc_reward(hd[:, 64:], z.to(dev), re, H * N, Z, 1.0)
        advs, rets = torch.zeros(H, N, 1, device=dev), torch.zeros(H, N, 1, device=dev)
        b.gae(dones.to(dev), val.to(dev), nval.to(dev), rt.to(dev), rd, re, 0.0, 0.5, 0.5, 0.99, 0.95, advs, rets, H, N)
        acc3 = torch.zeros(3, dtype=torch.float64, device=dev)
        adv = torch.zeros(H * N, device=dev)
        b.adv_norm(rets, val.to(dev), mask.to(dev), adv, acc3, H * N, 1, 0)
        b.adv_norm(rets, val.to(dev), mask.to(dev), adv, acc3, H * N, 1, 1)
        ring = torch.zeros(700, A, device=dev)
        b.ring_store(amp.to(dev), A, idx.to(dev), (H, N), 500, ring, 700, 450)
        outs.append((rd.cpu(), re.cpu(), advs.cpu(), rets.cpu(), adv.cpu(), ring.cpu()))
    names = ['disc_r', 'enc_r', 'advs', 'returns', 'adv_norm', 'ring']
    for a, c, nm in zip(outs[0], outs[1], names):
        close(a, c, 2e-5, 2e-5, nm)
    assert torch.equal(outs[0][5], outs[1][5])


def test_sample_latents(be):
    z = torch.zeros(20000, 64).cuda()
    st = torch.tensor([1234, 0], dtype=torch.int64).cuda()
    be.sample_latents(z, 20000, 64, st)
    z2 = torch.zeros(20000, 64).cuda()
    be.sample_latents(z2, 20000, 64, st)
    assert int(st[1]) == 2
    zc = z.cpu()
    close(zc.norm(dim=-1), torch.ones(20000), 1e-5, 1e-5, 'unit rows')
    assert not torch.equal(zc, z2.cpu())
    # isotropy: component mean ~ 0, second moment ~ 1/64
    assert float(zc.mean(0).abs().max()) < 0.01
    close((zc * zc).mean(0), torch.full((64,), 1 / 64), 0.08, 0.0, 'second moment')


def test_multi_ops(be):
    """Pointer-table launches (all layers' shadows / all minibatch fields in one kernel) == the per-item ops."""
    dev = 'cuda'
    g = torch.Generator().manual_seed(4)
    items, rows = [], []
    for n, k, ss, sd, kp in [(48, 53, 37, 64, 128), (1, 24, 24, 24, 64), (100, 1400, 1400, 1400, 1408)]:
        W = torch.randn(n, k, generator=g).to(dev)
        b = torch.randn(n, generator=g).to(dev)
        npad = (n + 63) // 64 * 64
        ws, wts = torch.zeros(npad, kp, dtype=torch.bfloat16, device=dev), torch.zeros(kp, npad, dtype=torch.bfloat16, device=dev)
        bs = torch.zeros(npad, device=dev)
        rows.append([W.data_ptr(), n, k, ws.data_ptr(), ws.stride(0), wts.data_ptr(), wts.stride(0), ss, sd - ss, b.data_ptr(),
                     bs.data_ptr(), (k + 31) // 32])
        items.append((W, ws, wts, ss, sd, b, bs))
    be.refresh_shadow_multi(torch.tensor(rows, dtype=torch.int64, device=dev), items, torch.bfloat16)
    for W, ws, wts, ss, sd, b, bs in items:
        rw, rwt = torch.zeros_like(ws), torch.zeros_like(wts)
        be.refresh_shadow(W, rw, rwt, ss, sd)
        assert torch.equal(ws, rw) and torch.equal(wts, rwt) and torch.equal(bs[:b.numel()], b)
    H, N, M = 8, 40, 123
    idx = torch.randperm(H * N, generator=g)[:M].to(torch.int32).to(dev)
    fields, rows = [], []
    for D in (31, 1, 64):
        src = torch.randn(H * N, D, generator=g).to(dev)
        dst = torch.zeros(M, D, device=dev)
        rows.append([src.data_ptr(), src.stride(0), D, dst.data_ptr(), dst.stride(0), L.F32])
        fields.append((src, D, dst))
    be.gather_multi(torch.tensor(rows, dtype=torch.int64, device=dev), fields, idx, (H, N), M)
    for src, D, dst in fields:
        ref = torch.zeros_like(dst)
        be.gather_rows(src, D, idx, (H, N), M, ref)
        assert torch.equal(dst, ref)


@pytest.mark.parametrize('dt', DT)
def test_stacked_rows_aux_wrap_and_bias_rows(be, dt):
    """Row blocks stacked under one launch: aux rows wrap back (m >= split reads aux[m - delta]); the bias gradient
    of the TN kernel only sums the first bias_rows rows."""
    M, N, K, split, delta = 512, 128, 256, 384, 128
    g = torch.Generator().manual_seed(31)
    A = (torch.randn(M, K, generator=g) * 0.3).to(dt)
    B = (torch.randn(N, K, generator=g) * 0.1).to(dt)
    aux = torch.randn(M, N, generator=g).to(dt)
    X = (torch.randn(M, 192, generator=g) * 0.3).to(dt)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        C = torch.zeros(M, N, dtype=dt, device=dev)
        b.gemm_nt(A.to(dev), B.to(dev), C, M, N, K, aux=aux.to(dev), aux_mode=L.AUX_RELU_MASK, aux_split=split, aux_delta=delta)
        G, gb = torch.zeros(N, 192, device=dev), torch.zeros(N, device=dev)
        b.gemm_tn(C, X.to(dev), G, M, N, 192, N, 192, 192, 192, gbias=gb, bias_rows=split)
        outs.append((C.float().cpu(), G.cpu(), gb.cpu()))
    rt, at = _tol(dt)
    close(outs[0][0], outs[1][0], rt, at * 2, 'C')
    close(outs[0][1], outs[1][1], 1e-3 if dt != torch.float32 else 3e-5, 2e-3, 'G')
    close(outs[0][2], outs[1][2], 1e-3 if dt != torch.float32 else 3e-5, 2e-3, 'gbias')


@pytest.mark.parametrize('M,N,K', [(513, 320, 1408), (1024, 1024, 1024), (77, 64, 128), (2048, 1024, 320)])
def test_gemm_x3_split_mode(M, N, K):
    """ASE_F32X3: f32 storage, products as three bf16 MFMAs on a hi/lo split: results within ~1e-5 of exact f32
    (relative to the row/column scale), forward-type and weight-gradient-type."""
    from ase_amd.backend import HipBackend
    b3 = HipBackend(x3=True)
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g) * 0.5
    B = torch.randn(N, K, generator=g) * 0.1
    bias = torch.randn(N, generator=g)
    Cg = torch.zeros(M, N).cuda()
    b3.gemm_nt(A.cuda(), B.cuda(), Cg, M, N, K, bias=bias.cuda(), act=L.ACT_NONE)
    ref = A.double() @ B.double().t() + bias.double()
    scale = float((A.abs().double() @ B.abs().double().t()).max())
    assert float((Cg.cpu().double() - ref).abs().max()) <= 3e-5 * scale
    X = torch.randn(M, 192, generator=g) * 0.3
    G = torch.zeros(N, 192).cuda()
    b3.gemm_tn(Cg, X.cuda(), G, M, N, 192, N, 192, 192, 192)
    gref = Cg.cpu().double().t() @ X.double()
    gscale = float((Cg.cpu().abs().double().t() @ X.abs().double()).max())
    assert float((G.cpu().double() - gref).abs().max()) <= 3e-5 * gscale


@pytest.mark.parametrize('M,N,K,ea,eb,sa,sb', [(513, 320, 1408, 12, 11, 1.5, 0.03), (4096, 1024, 1408, 12, 11, 1.5, 0.03),
                                                (4096, 512, 1024, 6, 11, 0.7, 0.03), (4096, 1408, 1024, 12, 11, 0.02, 0.03),
                                                (77, 64, 128, 6, 11, 30.0, 0.5), (1024, 1024, 320, 12, 11, 1e-3, 1e-3),
                                                (512, 256, 128, 6, 11, 1e-4, 0.03)])
def test_gemm_x3_half_split(M, N, K, ea, eb, sa, sb):
    """ASE_F32H3 (the gradient penalty's value path in 'f16gpx3'): f32 storage, every product as three f16 MFMAs on hi / lo splits
    of operands scaled by 2^ea / 2^eb.  Against the emulator's restatement of the same split (f64 accumulation) and against the
    exact product: ~22 significant bits per operand - 1e-6 of the row/column scale where the bf16 split (ASE_F32X3) holds 3e-5.
    Cases 5 and 7 sit at the edges of half's range (large activations; tiny operands whose lo parts are SUBNORMAL halves - the
    matrix cores of gfx950 do not flush them, which the bit-level comparison with the emulator pins)."""
    from ase_amd.backend import HipBackend
    from tests.emu_backend import x3_half_product
    bh = HipBackend(x3='f16')
    g = torch.Generator().manual_seed(M + N + ea)
    A = torch.randn(M, K, generator=g) * sa
    A[:, ::7] = 0.0                                            # ReLU-style exact zeros
    B = torch.randn(N, K, generator=g) * sb
    bias = torch.randn(N, generator=g)
    Cg = torch.zeros(M, N).cuda()
    bits = torch.zeros(M, (N + 31) // 32, dtype=torch.int32).cuda()
    relu = N % 32 == 0
    # B in the packed split format the shadow refresh writes (and its transpose: checked through the unpacker)
    Bs, Bts = torch.zeros(N, K).cuda(), torch.zeros(K, (N + 7) // 8 * 8).cuda()
    bh.refresh_shadow(B.cuda(), Bs, Bts, K, K, x3_exp=eb)
    from tests.emu_backend import half_split, unpack_split
    hi, lo = half_split(B, eb)
    want = (hi + lo) * 2.0 ** -eb
    assert torch.equal(unpack_split(Bs.cpu(), N, K, eb), want) and torch.equal(unpack_split(Bts.cpu(), K, N, eb), want.t())
    bh.gemm_nt(A.cuda(), Bs, Cg, M, N, K, bias=bias.cuda(), act=L.ACT_RELU if relu else L.ACT_NONE,
               mask_out=bits if relu else None, x3_exps=(ea, eb))
    pre = A.double() @ B.double().t() + bias.double()
    ref = pre.clamp_min(0) if relu else pre
    emu = x3_half_product(A, B, ea, eb).double() + bias.double()
    emu = emu.clamp_min(0) if relu else emu
    scale = float((A.abs().double() @ B.abs().double().t()).max()) + float(bias.abs().max())
    got = Cg.cpu().double()
    assert float((got - emu).abs().max()) <= 5e-7 * scale, float((got - emu).abs().max()) / scale      # f32 accumulation order only
    tol = 1e-6 if sa * 2.0 ** ea >= 1.0 else 2e-5             # (tiny operands: lo parts subnormal, documented degradation)
    assert float((got - ref).abs().max()) <= tol * scale, float((got - ref).abs().max()) / scale
    if relu:                                                   # the mask twin is the sign of what was stored
        w = bits.cpu().to(torch.int64) & 0xFFFFFFFF
        mb = ((w.unsqueeze(-1) >> torch.arange(32)) & 1).reshape(M, -1)[:, :N].bool()
        assert bool((mb == (got > 0)).all())
    # the same launch in the bf16 split is several times further from the exact product (why the value path moved; the
    # half split's own error sits near the f32 accumulation noise of the long sums)
    b3 = HipBackend(x3=True)
    C3 = torch.zeros(M, N).cuda()
    b3.gemm_nt(A.cuda(), B.cuda(), C3, M, N, K, bias=bias.cuda(), act=L.ACT_RELU if relu else L.ACT_NONE)
    if K >= 1024 and sa >= 0.5:        # (long sums of O(1) terms: elsewhere both errors sit in the f32 rounding of the bias add)
        assert float((C3.cpu().double() - ref).abs().max()) >= 3 * float((got - ref).abs().max())


def test_gemm_x3_half_split_rejects_bad_scales():
    from ase_amd.backend import HipBackend
    bh = HipBackend(x3='f16')
    A, B, C_ = torch.zeros(64, 64).cuda(), torch.zeros(64, 64).cuda(), torch.zeros(64, 64).cuda()
    with pytest.raises(L.AseHipError):
        bh.gemm_nt(A, B, C_, 64, 64, 64, x3_exps=(30, 11))
    with pytest.raises(L.AseHipError):
        bh.refresh_shadow(A, B, None, 64, 64, x3_exp=30)
    # an operand beyond half's range is LOUD: the output turns NaN instead of carrying a silently saturated product
    A[3, 5] = 3000.0
    bh.refresh_shadow(torch.ones(64, 64).cuda(), B, None, 64, 64, x3_exp=11)
    bh.gemm_nt(A, B, C_, 64, 64, 64, x3_exps=(12, 11))
    assert bool(torch.isnan(C_[3]).all()) and bool(torch.isfinite(C_[4]).all())


@pytest.mark.parametrize('dt', DT)
@pytest.mark.parametrize('M,N,K', [(300, 192, 256), (777, 64, 128), (1024, 1024, 1024), (4096, 512, 320), (16384, 1024, 256)])
def test_gemm_nt_relu_bit_mask(be, dt, M, N, K):
    """mask_out of a forward ReLU layer = (stored activation > 0), bit-exact against the emulation; the data-gradient
    launch that reads the bits (ASE_AUX_RELU_BITS, with a stacked row block wrapping onto earlier rows) equals the one
    that reads the activation itself (ASE_AUX_RELU_MASK), bit for bit."""
    g = torch.Generator().manual_seed(M + N)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dt)
    B = (torch.randn(N, K, generator=g) * 0.1).to(dt)
    bias = torch.randn(N, generator=g)
    Hg, Hc = torch.zeros(M, N, dtype=dt).cuda(), torch.zeros(M, N, dtype=dt)
    Wg = torch.full((M, N // 32 + 2), -1, dtype=torch.int32).cuda()        # ldmask > N / 32
    Wc = torch.full((M, N // 32 + 2), -1, dtype=torch.int32)
    be.gemm_nt(A.cuda(), B.cuda(), Hg, M, N, K, bias=bias.cuda(), act=L.ACT_RELU, mask_out=Wg)
    EmuBackend().gemm_nt(A, B, Hc, M, N, K, bias=bias, act=L.ACT_RELU, mask_out=Wc)
    w = Wg.cpu().to(torch.int64) & 0xFFFFFFFF
    bits = ((w[:, :N // 32].unsqueeze(-1) >> torch.arange(32)) & 1).reshape(M, N).bool()
    assert torch.equal(bits, Hg.float().cpu() > 0)
    assert torch.equal(Wg.cpu()[:, N // 32:], Wc[:, N // 32:])             # pad words untouched
    # data gradient with 1.5 M rows: the extra half block re-uses the masks of rows M/2 .. M
    # (the last shape is a whole number of rounds of 256 x 256 tiles in both launches: the phased bf16 kernel)
    K2, R = 128, (2 * M if M >= 16384 else M + M // 2)
    dY = (torch.randn(R, K2, generator=g) * 0.3).to(dt).cuda()
    Wt = (torch.randn(N, K2, generator=g) * 0.1).to(dt).cuda()
    d1, d2 = torch.zeros(R, N, dtype=dt).cuda(), torch.zeros(R, N, dtype=dt).cuda()
    delta = R - M
    be.gemm_nt(dY, Wt, d1, R, N, K2, aux=Hg, aux_mode=L.AUX_RELU_MASK, aux_split=M, aux_delta=delta)
    be.gemm_nt(dY, Wt, d2, R, N, K2, aux=Wg, aux_mode=L.AUX_RELU_BITS, aux_split=M, aux_delta=delta)
    assert torch.equal(d1, d2)
    assert float(d1.float().abs().max()) > 0


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
def test_gemm_tn_grouped_matches_per_layer(be, dt):
    """One grouped launch over several layers (different row counts, widths, concat column maps, bias row limits)
    against the per-layer kernel and the emulation."""
    g = torch.Generator().manual_seed(21)
    shapes = [(2048, 512, 1024, 512, 1024, 1024, 1024, 0), (4096, 1024, 320, 1024, 317, 253, 256, 0),
              (1024, 256, 1408, 200, 1400, 1400, 1400, 768), (2048, 1024, 1024, 1024, 1024, 1024, 1024, 1536)]
    probs, ref = [], []
    for (M, N, K, nr, kr, ss, sd, br) in shapes:
        A = (torch.randn(M, N, generator=g) * 0.2).to(dt)
        B = (torch.randn(M, K, generator=g) * 0.2).to(dt)
        G0, b0 = torch.randn(nr, kr, generator=g), torch.randn(nr, generator=g)
        Gc, bc = G0.clone(), b0.clone()
        EmuBackend().gemm_tn(A, B, Gc, M, N, K, nr, kr, ss, sd, alpha=0.5, gbias=bc, bias_rows=br)
        ref.append((Gc, bc, M))
        probs.append((A.cuda(), B.cuda(), G0.cuda(), b0.cuda(), br if br else M, M, N, K, nr, kr, ss, sd, 0.5))
        assert be.grouped_tn_ok(dt, M, nr, K, br if br else M)
    plan = be.make_tn_plan(probs)
    assert plan['n_work'] >= len(shapes)
    be.gemm_tn_grouped(plan)
    for (A, B, G, gb, *_), (Gc, bc, M) in zip(probs, ref):
        close(G, Gc, 3e-5, 3e-5 * math.sqrt(M), 'grouped tn G')
        close(gb, bc, 3e-5, 3e-5 * math.sqrt(M), 'grouped tn bias')


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('M,N,K', [(16384, 1024, 1024), (8192, 512, 1408), (16000, 1000, 192), (32768, 256, 64)])
def test_gemm_nt_phased_tile(be, dt, M, N, K):
    """Shapes that take the phased 256 x 256 kernel (whole rounds of tiles) incl. ragged edges and 1-3 K-tiles."""
    g = torch.Generator().manual_seed(K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dt)
    B = (torch.randn(N, K, generator=g) * 0.1).to(dt)
    bias = torch.randn(N, generator=g)
    Cg, Cc = torch.zeros(M, N, dtype=dt).cuda(), torch.zeros(M, N, dtype=dt)
    be.gemm_nt(A.cuda(), B.cuda(), Cg, M, N, K, bias=bias.cuda(), act=L.ACT_RELU)
    EmuBackend().gemm_nt(A, B, Cc, M, N, K, bias=bias, act=L.ACT_RELU)
    rt, at = _tol(dt)
    close(Cg.float(), Cc.float(), rt, at * math.sqrt(K / 64), f'nt phased {M}x{N}x{K}')


@pytest.mark.parametrize('dt', [torch.bfloat16, torch.float16])
@pytest.mark.parametrize('M,N,K', [(12288, 1024, 1408), (12288, 1024, 64), (12200, 1024, 192), (6144, 2048, 512)])
def test_gemm_nt_phased_tile_192_rows(be, dt, M, N, K):
    """Row counts whose 256-row tiling leaves CUs idle in a single round take the 192 x 256 variant of the phased kernel
    (kernel id 6; 12288 rows = the discriminator's three 4096-row blocks): forward ReLU with the bit-mask twin, then the
    data-gradient launch reading those bits with a wrapped row block, incl. a ragged last row tile and 1 / 3 / 8 / 22 K-tiles."""
    assert be.lib.ase_hip_gemm_nt_kernel_id(M, N, K, be._gemm_code(dt)) == 6
    g = torch.Generator().manual_seed(K)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dt)
    B = (torch.randn(N, K, generator=g) * 0.1).to(dt)
    bias = torch.randn(N, generator=g)
    Cg, Cc = torch.zeros(M, N, dtype=dt).cuda(), torch.zeros(M, N, dtype=dt)
    Wg, Wc = torch.zeros(M, N // 32, dtype=torch.int32).cuda(), torch.zeros(M, N // 32, dtype=torch.int32)
    be.gemm_nt(A.cuda(), B.cuda(), Cg, M, N, K, bias=bias.cuda(), act=L.ACT_RELU, mask_out=Wg)
    EmuBackend().gemm_nt(A, B, Cc, M, N, K, bias=bias, act=L.ACT_RELU, mask_out=Wc)
    rt, at = _tol(dt)
    close(Cg.float(), Cc.float(), rt, at * math.sqrt(K / 64), f'nt 192-row tile {M}x{N}x{K}')
    w = Wg.cpu().to(torch.int64) & 0xFFFFFFFF
    bits = ((w.unsqueeze(-1) >> torch.arange(32)) & 1).reshape(M, N).bool()
    assert torch.equal(bits, Cg.float().cpu() > 0)
    # data gradient: rows [M/2, M) read the masks of rows [0, M/2) again (aux_split / aux_delta), bits == activation masks
    split = M // 2
    dY = (torch.randn(M, K, generator=g) * 0.3).to(dt).cuda()
    Wt = (torch.randn(N, K, generator=g) * 0.1).to(dt).cuda()
    d1, d2 = torch.zeros(M, N, dtype=dt).cuda(), torch.zeros(M, N, dtype=dt).cuda()
    be.gemm_nt(dY, Wt, d1, M, N, K, aux=Cg, aux_mode=L.AUX_RELU_MASK, aux_split=split, aux_delta=split)
    be.gemm_nt(dY, Wt, d2, M, N, K, aux=Wg, aux_mode=L.AUX_RELU_BITS, aux_split=split, aux_delta=split)
    assert torch.equal(d1, d2)
    ref = (dY.float().cpu() @ Wt.float().cpu().t()) * (Cg.float().cpu() > 0).float()[torch.cat([torch.arange(split), torch.arange(M - split)])]
    close(d2.float().cpu(), ref.to(dt).float(), rt, at * math.sqrt(K / 64), 'nt 192-row tile data gradient')


@pytest.mark.parametrize('dt', DT)
def test_apply_multi_fused_optimizer_step(be, dt):
    """ase_hip_apply_multi = weight-only gradient terms + their norms + Adam + shadow refresh, against the separate ops
    of the emulation (axpy / reduce_sum / adam / refresh_shadow) on two layers with concat column maps."""
    import struct
    g = torch.Generator().manual_seed(7)
    # (n, k, split_src, split_dst, coefficient, slot a, slot b, wide): wide = the kernel's 16-byte path (k, split and gap in
    # whole 4-element chunks); its parameter / gradient / moment tensors start at an odd 4-byte offset of their buffers
    specs = [(48, 53, 37, 64, 0.02, 3, 5, 0), (130, 200, 200, 200, 0.0, -1, -1, 0), (1, 70, 70, 70, 0.5, 4, -1, 0),
             (130, 200, 200, 200, 0.0, -1, -1, 1), (300, 520, 256, 264, 0.01, 6, 7, 1), (31, 512, 512, 512, 0.0, -1, -1, 1)]
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        gg = torch.Generator().manual_seed(7)
        rows, items, keep = [], [], []
        acc = torch.zeros(8, dtype=torch.float64, device=dev)
        st = torch.tensor([3.0, 1e-3, 0.9, 0.999, 1e-8, 1 - 0.9 ** 3, 1 - 0.999 ** 3, 0.0], dtype=torch.float64, device=dev)
        for (n, k, ss, sd, coef, sa, sb, wide) in specs:
            odd = lambda t: torch.cat([t.new_zeros(1), t.reshape(-1)])[1:].view_as(t)       # 4-byte aligned only
            W, b_ = odd(torch.randn(n, k, generator=gg).to(dev)), torch.randn(n, generator=gg).to(dev)
            gW, gb = odd(torch.randn(n, k, generator=gg).to(dev)), torch.randn(n, generator=gg).to(dev)
            mW, vW = odd(torch.randn(n, k, generator=gg).to(dev) * 0.1), odd(torch.rand(n, k, generator=gg).to(dev) * 0.01)
            mb, vb = torch.randn(n, generator=gg).to(dev) * 0.1, torch.rand(n, generator=gg).to(dev) * 0.01
            npad, kpad = (n + 63) // 64 * 64, (k + (sd - ss) + 63) // 64 * 64
            Ws, Wts = torch.zeros(npad, kpad, dtype=dt, device=dev), torch.zeros(kpad, npad, dtype=dt, device=dev)
            bs = torch.zeros(npad, device=dev)
            rows.append([W.data_ptr(), n, k, Ws.data_ptr(), Ws.stride(0), Wts.data_ptr(), Wts.stride(0), ss, sd - ss,
                         b_.data_ptr(), bs.data_ptr(), (k + 31) // 32, gW.data_ptr(), mW.data_ptr(), vW.data_ptr(),
                         gb.data_ptr(), mb.data_ptr(), vb.data_ptr(), struct.unpack('<i', struct.pack('<f', coef))[0], sa, sb, wide, 0, 0])
            items.append((W, Ws, Wts, ss, sd, b_, bs[:n], gW, mW, vW, gb, mb, vb, coef, sa, sb))
            keep.append((W, b_, gW, mW, vW, mb, vb, Ws, Wts, bs))
        desc = torch.tensor(rows, dtype=torch.int64, device=dev)
        b.apply_multi(desc, items, dt, st, acc)
        outs.append(([tuple(t.float().cpu() for t in kk) for kk in keep], acc.cpu()))
    for kg, kc in zip(outs[0][0], outs[1][0]):
        for i, (a, c) in enumerate(zip(kg, kc)):
            if i >= 7:      # shadows / bias copy: the updated weights differ by f32 rounding noise between the two Adam codes
                close(a, c, 1e-6 if dt == torch.float32 else 8e-3, 1e-7, f'shadow {i}')
            else:
                close(a, c, 1e-6, 1e-7, f'apply_multi tensor {i}')
    close(outs[0][1], outs[1][1], 1e-9, 1e-12, 'norm accumulators')


@pytest.mark.parametrize('local_root,root_h', [(True, True), (True, False), (False, True), (False, False)])
def test_build_amp_obs_matches_reference(be, local_root, root_h, golden_dir):
    """N2: ase_hip_build_amp_obs against the reference's build_amp_observations (golden) and its history update."""
    import os
    from oracle import amp_obs as A
    G = torch.load(os.path.join(golden_dir, 'amp_obs.pt'), weights_only=False)
    i = {k: v.contiguous().cuda() for k, v in G['inputs'].items()}
    ref = G['outputs'][(local_root, root_h)]
    N, F, S = ref.shape[0], ref.shape[1], 10
    g = torch.Generator().manual_seed(9)
    hist0 = torch.randn(N, S, F, generator=g)
    hist = hist0.clone().cuda()
    be.build_amp_obs(i['root_pos'], i['root_rot'], i['root_vel'], i['root_ang_vel'], i['dof_pos'], i['dof_vel'],
                     i['key_body_pos'], G['dof_offsets'], local_root, root_h, hist, shift=True)
    want = A.push_history(hist0.clone(), ref)
    close(hist[:, 0].cpu(), ref, 1e-5, 3e-6, 'amp obs frame')
    assert torch.equal(hist[:, 1:].cpu(), want[:, 1:])                  # the shifted past is a pure copy
    # without the shift only slot 0 changes; a ragged environment count exercises the partial workgroup
    M = 70
    hist = hist0[:M].clone().cuda()
    be.build_amp_obs(*[i[k][:M].contiguous() for k in ('root_pos', 'root_rot', 'root_vel', 'root_ang_vel', 'dof_pos', 'dof_vel',
                                                      'key_body_pos')], G['dof_offsets'], local_root, root_h, hist, shift=False)
    close(hist[:, 0].cpu(), ref[:M], 1e-5, 3e-6, 'amp obs frame (no shift)')
    assert torch.equal(hist[:, 1:].cpu(), hist0[:M, 1:])


def test_motion_state_matches_reference(be, golden_dir):
    """N2: ase_hip_motion_state against the reference's MotionLib.get_motion_state (golden, two shipped clips), then the
    demo-side chain motion state -> build_amp_obs against the oracle chain."""
    import os
    from oracle import amp_obs as A
    G = torch.load(os.path.join(golden_dir, 'motion_state.pt'), weights_only=False)
    c = G['clips']
    dev = {k: c[k].float().contiguous().cuda() for k in ('gts', 'grs', 'lrs', 'grvs', 'gravs', 'dvs', 'lengths', 'dt')}
    dev.update({k: c[k].to(torch.int32).cuda() for k in ('num_frames', 'length_starts')})
    dev.update({k: c[k] for k in ('dof_body_ids', 'dof_offsets', 'key_body_ids')})
    ids, t = G['motion_ids'].to(torch.int32).cuda(), G['times'].float().cuda()
    out = be.motion_state(dev, ids, t)
    names = list(G['outputs'])
    for k, o in zip(names, out):
        ref = G['outputs'][k]
        # interpolated rotations go through acos / sin on both sides (device vs host libm): 1e-5; copies are exact
        if k in ('root_vel', 'root_ang_vel', 'dof_vel'):
            assert torch.equal(o.cpu(), ref), k
        else:
            close(o.cpu(), ref, 2e-5, 2e-5, 'motion ' + k)
    n = ids.shape[0]
    hist = torch.zeros(n, 1, 13 + 6 * 13 + 31 + 18, device='cuda')
    be.build_amp_obs(*out[:2], out[3], out[4], out[2], out[5], out[6], c['dof_offsets'], True, True, hist, shift=False)
    o = G['outputs']
    want = A.build_amp_observations(o['root_pos'], o['root_rot'], o['root_vel'], o['root_ang_vel'], o['dof_pos'], o['dof_vel'],
                                    o['key_pos'], True, True, c['dof_offsets'])
    close(hist[:, 0].cpu(), want, 1e-4, 1e-4, 'demo amp obs')


# ------------------------------------------------------------------------------------------------ round-2 operators
def test_sample_latents_rank_offsets(be):
    """Rows drawn with row_offset = the global index of a rank's first row equal the corresponding rows of the one-rank draw
    (data-parallel ranks draw disjoint parts of ONE stream); advance=False leaves the stream position alone; the second
    output is the compute-dtype copy."""
    st = torch.tensor([99, 5], dtype=torch.int64).cuda()
    full = torch.zeros(4096, 64).cuda()
    be.sample_latents(full, 4096, 64, st, advance=False)
    assert int(st[1]) == 5
    part = torch.zeros(1024, 64).cuda()
    copy = torch.zeros(1024, 64, dtype=torch.bfloat16).cuda()
    be.sample_latents(part, 1024, 64, st, row_offset=2048, advance=False, z2=copy)
    assert torch.equal(part, full[2048:3072])
    assert torch.equal(copy, part.to(torch.bfloat16))


def test_sample_actions(be):
    """ase_hip_sample_actions: mu / sigma passthrough (tanh option), Normal sample statistics, the reference's neglogp of
    the SAMPLED action, eps-greedy substitution and mask frequencies (learning/amp_agent.py:139-169)."""
    import math
    n, A, ld = 50000, 31, 64
    g = torch.Generator().manual_seed(0)
    mu = (torch.randn(n, ld, generator=g) * 0.3).cuda()
    logstd = torch.full((A,), -2.9).cuda()
    probs = torch.rand(n, generator=g).cuda()
    probs[:100] = 1.0
    probs[100:200] = 0.0
    st = torch.tensor([7, 0], dtype=torch.int64).cuda()
    out = {k: torch.zeros(n, A).cuda() for k in ('mu', 'sigma', 'act')}
    nlp, mask = torch.zeros(n).cuda(), torch.zeros(n).cuda()
    be.sample_actions(mu, logstd, probs, st, out['mu'], out['sigma'], out['act'], nlp, mask, n, A, mu_tanh=True)
    assert int(st[1]) == 1
    m = torch.tanh(mu[:, :A])
    close(out['mu'], m, 1e-6, 1e-6, 'mu')
    close(out['sigma'], torch.exp(logstd).expand(n, A), 1e-6, 0, 'sigma')
    assert bool((mask[:100] == 1).all()) and bool((mask[100:200] == 0).all())
    assert abs(float(mask.mean()) - float(probs.mean())) < 0.01                       # Bernoulli(p_row)
    det = mask == 0
    assert torch.equal(out['act'][det], out['mu'][det])
    eps = (out['act'][~det] - m[~det]) / math.exp(-2.9)
    assert abs(float(eps.mean())) < 0.01 and abs(float(eps.std()) - 1.0) < 0.01       # N(0, 1) noise
    ref = 0.5 * (eps ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * A + A * (-2.9)
    close(nlp[~det], ref, 1e-4, 1e-3, 'neglogp of the sampled action')
    assert bool(torch.isfinite(nlp).all()) and float(nlp[det].std()) > 0              # deterministic rows keep the sample's neglogp


def test_normalize_rows_and_clip_scale(be):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1000, 64, generator=g).cuda()
    x[3] = 0
    y = torch.zeros_like(x)
    be.normalize_rows(x, y, 1000, 64)
    close(y, torch.nn.functional.normalize(x, dim=-1), 1e-6, 1e-6, 'normalize_rows')
    grad = torch.randn(100000, generator=g).cuda()
    acc = torch.zeros(24, dtype=torch.float64).cuda()
    be.reduce_sum(grad, grad.numel(), True, acc, 17)
    ref = grad * min(1.0, 3.0 / (float(grad.norm()) + 1e-6))
    be.clip_scale(grad, acc, 17, 3.0)
    close(grad, ref, 1e-5, 1e-6, 'clip_scale')
    big = grad.clone()
    be.clip_scale(big, acc, 17, 1e9)                                                   # norm below the bound: unchanged
    assert torch.equal(big, grad)


def test_rms_multi_matches_single(be):
    """The 3-stream moments / normalise launches equal three single-stream launches (same per-workgroup partial sums; the
    f64 atomics that combine the workgroups arrive in any order, hence equality to rounding of the f64 sums)."""
    g = torch.Generator().manual_seed(2)
    D, M = 1400, 4096
    srcs = [(torch.randn(9000, D, generator=g) * (1 + s) + s).cuda() for s in range(3)]
    idxs = [torch.randint(0, 9000, (M,), generator=g).to(torch.int32).cuda() for _ in range(3)]
    state = torch.zeros(2 * D + 1, dtype=torch.float64).cuda()
    state[D:] = 1.0
    streams = [(srcs[s], idxs[s], (0, 0)) for s in range(3)]
    a = torch.zeros(3, 2 * D, dtype=torch.float64).cuda()
    b = torch.zeros(3, 2 * D, dtype=torch.float64).cuda()
    be.rms_moments_multi(streams, D, M, state, [a[s] for s in range(3)])
    for s in range(3):
        be.rms_moments(srcs[s], D, idxs[s], (0, 0), M, state, b[s])
    close(a, b, 1e-12, 1e-9, 'multi vs single moments')
    ref = srcs[1][idxs[1].long()].double()
    close(a[1][:D], ref.sum(0), 1e-9, 1e-6, 'sum')
    mean, std = torch.zeros(3, D).cuda(), torch.zeros(3, D).cuda()
    be.rms_finalize(state, D, a, M, 3, mean, std)
    outs = [torch.zeros(M, 1408, dtype=torch.bfloat16).cuda() for _ in range(3)]
    outs1 = [torch.zeros(M, 1408, dtype=torch.bfloat16).cuda() for _ in range(3)]
    be.rms_normalize_multi(streams, D, M, [mean[s] for s in range(3)], [std[s] for s in range(3)], outs)
    for s in range(3):
        be.rms_normalize(srcs[s], D, idxs[s], (0, 0), M, mean[s], std[s], [outs1[s]])
        assert torch.equal(outs[s], outs1[s])          # (same means / stds in: identical outputs)


def test_launch_program_replay(be):
    """A recorded launch program replays the same launches on the same streams (fork / join included) with new data in the
    same buffers; nothing runs while recording."""
    x = torch.randn(4096, 64).cuda()
    y = torch.zeros(4096, 64).cuda()
    w = torch.zeros(4096, 64).cuda()
    side = torch.cuda.Stream()
    prog = be.prog_create()
    be.prog_begin(prog)
    fork = be.mark()
    with torch.cuda.stream(side):
        be.wait(fork)
        be.normalize_rows(x, y, 4096, 64)
        done = be.mark()
    be.zero_(w)
    be.wait(done)
    be.copy_(w, y)
    be.prog_end(prog)
    torch.cuda.synchronize()
    assert float(y.abs().sum()) == 0 and be.prog_size(prog) == 7            # recorded, not executed
    for _ in range(3):
        x.normal_()
        torch.cuda.synchronize()
        be.prog_launch(prog)
        torch.cuda.synchronize()
        assert torch.allclose(w, torch.nn.functional.normalize(x, dim=-1), rtol=1e-6, atol=1e-6)
    be.prog_destroy(prog)


def test_launch_program_keeps_dependencies_and_callback_order(be):
    """A chain of copies handed from stream to stream through marks (four streams, one stage recorded before the stage it depends on is
    complete in program order) - a wait that does not see its record reads the previous replay's data - and host callbacks recorded on
    different streams run in PROGRAM order (the collectives of a data-parallel step).  400 replays with new data each."""
    n = 1 << 22
    src, a, b, c, out = (torch.zeros(n, device='cuda') for _ in range(5))
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    order = []
    prog = be.prog_create()
    be.prog_begin(prog)
    fork = be.mark()
    with torch.cuda.stream(s1):
        be.wait(fork)
        be.host_call(lambda: order.append(0))
        be.copy_(a, src)
        m1 = be.mark()
    with torch.cuda.stream(s3):                       # (recorded BEFORE the stage it depends on is complete in program order)
        be.wait(fork)
        be.zero_(c)
        m0 = be.mark()
    with torch.cuda.stream(s2):
        be.wait(m1)
        be.copy_(b, a)
        be.host_call(lambda: order.append(1))
        m2 = be.mark()
    with torch.cuda.stream(s3):
        be.wait(m2)
        be.wait(m0)
        be.copy_(c, b)
        m3 = be.mark()
    be.host_call(lambda: order.append(2))
    be.wait(m3)
    be.copy_(out, c)
    be.prog_end(prog)
    assert not order
    for k in range(1, 401):
        src.fill_(float(k))
        torch.cuda.synchronize()
        be.prog_launch(prog)
        torch.cuda.synchronize()
        assert float(out[0]) == k and float(out[-1]) == k and float(out.min()) == k, (k, float(out[0]), float(out[-1]))
        assert order == [0, 1, 2], (k, order)
        order.clear()
    be.prog_destroy(prog)


@pytest.mark.parametrize('kl,expect', [(0.05, 2e-5 / 1.5), (0.001, 2e-5 * 1.5), (0.01, 2e-5)])
def test_finalize_scalars_adaptive_lr(be, kl, expect):
    """The rl_games AdaptiveScheduler branch of ase_hip_finalize_scalars (learning/common_agent.py:204-208): lr / 1.5 when the
    step's kl exceeds 2 x kl_threshold, x 1.5 below half of it, unchanged between - device result = emulator = closed form.
    (First run on hardware: round 3's driver, green.)"""
    cfg = dict(critic_coef=5, entropy_coef=0.0, bounds_loss_coef=10, disc_coef=5, disc_logit_reg=0.01,
               disc_grad_penalty=5, disc_weight_decay=1e-4, enc_coef=5, enc_weight_decay=0.0, amp_diversity_bonus=0.01,
               enc_grad_penalty=0.0)
    outs = []
    for dev in ('cuda', 'cpu'):
        b = be if dev == 'cuda' else EmuBackend()
        acc = (torch.arange(L.ACC_COUNT, dtype=torch.float64) + 1.5).to(dev)
        acc[L.ACC_KL] = kl * 1000                              # kl = acc / m_global
        res = torch.zeros(L.RES_COUNT, device=dev)
        st = torch.tensor([0.0, 2e-5, 0.9, 0.999, 1e-8, 1.0, 1.0, 0.0], dtype=torch.float64, device=dev)
        b.finalize_scalars(acc, res, 1000, 250, 1, 1, 1, 1, cfg, opt_state=st, kl_threshold=0.008)
        outs.append((res.cpu(), st.cpu()))
    close(outs[0][0], outs[1][0], 1e-6, 1e-7, 'res')
    assert abs(float(outs[0][1][1]) - expect) <= 1e-12 and abs(float(outs[1][1][1]) - expect) <= 1e-12, (outs[0][1], outs[1][1])


def test_gp_second_saturated_units_stay_finite(be):
    """ase_hip_gp_second at |z| of a few tens: act' is a tiny normal number, act'^2 underflows - the term is evaluated as
    (act'' / act') (g / act') dg and must stay finite (round 3's advisor: act'' / act'^2 was inf there) and equal the emulator's."""
    rows, width = 64, 128
    g0 = torch.Generator().manual_seed(3)
    z = torch.randn(rows, width, generator=g0) * 3
    z[:8] = torch.linspace(-80.0, 80.0, 8 * width).view(8, width)          # saturated rows
    u = torch.randn(rows, width, generator=g0) * 0.1
    r = torch.randn(rows, width, generator=g0) * 0.1
    for act in (L.ACT_SIGMOID, L.ACT_ELU, L.ACT_SELU, L.ACT_GELU, L.ACT_SILU, L.ACT_SOFTPLUS, L.ACT_TANH):
        from tests.emu_backend import _twin_factors
        twin = torch.tanh(z) if act == L.ACT_TANH else z
        d1, _ = _twin_factors(act, twin)
        gv, dg = d1 * u, d1 * r
        outs = []
        for dev in ('cuda', 'cpu'):
            b = be if dev == 'cuda' else EmuBackend()
            dz = torch.zeros(rows, width, device=dev)
            b.gp_second(twin.to(dev), gv.to(dev), dg.to(dev), dz, rows, width, act)
            outs.append(dz.cpu())
        assert bool(torch.isfinite(outs[0]).all()) and bool(torch.isfinite(outs[1]).all()), act
        close(outs[0], outs[1], 1e-3, 1e-5 * float(outs[1].abs().max()) + 1e-9, f'gp_second act {act}')     # (device erf / exp vs torch's)
        # a NaN / inf that ARRIVES in the chain values (a diverging penalty) is passed on, not zeroed (round 4's advisor)
        gv2 = gv.clone()
        gv2[20, 5], gv2[21, 6] = float('nan'), float('inf')
        for dev in ('cuda', 'cpu'):
            b = be if dev == 'cuda' else EmuBackend()
            dz = torch.zeros(rows, width, device=dev)
            b.gp_second(twin.to(dev), gv2.to(dev), dg.to(dev), dz, rows, width, act)
            assert bool(torch.isnan(dz[20, 5])) and not bool(torch.isfinite(dz[21, 6])), (act, dev, dz[20, 5], dz[21, 6])
