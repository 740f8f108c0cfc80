"""GPU tests of the dynamic loss scale (csrc/scaler.hip behind config loss_scale: 'dynamic' - torch.cuda.amp.GradScaler's
found_inf / skipped step of the reference's mixed_precision path, learning/ase_agent.py:271-288): the two entry points against
the emulator's semantics (incl. the per-step backoff / growth of the device-resident scale, ABI 6, and the producers' own overflow
reports into their scale records, ABI 7), and the engine's skip -> backoff
-> clean step -> growth cycle against the static-scale engine
(tests/test_scaler_emu.py::check_dynamic_loss_scale, the same check the emulator passes on the CPU)."""
import os

import pytest
import torch

from tests.emu_backend import EmuBackend
from tests.test_scaler_emu import check_dynamic_loss_scale

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
    from ase_amd.backend import HipBackend
    return HipBackend()


@pytest.mark.parametrize('dtype', [torch.float16, torch.bfloat16, torch.float32])
def test_scaler_check_finds_every_overflow(be, dtype):
    """ase_hip_scaler_check over buffers of every alignment / length class (16-byte body, scalar head and tail): a single bad
    element anywhere is found, clean buffers - including half's largest unsaturated value and f32's largest finite one - are not."""
    dev = 'cuda'
    g = torch.Generator().manual_seed(3)
    base = (torch.rand(70000, generator=g) * 2 - 1).to(dtype).to(dev)
    big = {torch.float16: 65472.0, torch.bfloat16: 3.0e38, torch.float32: 3.4e38}[dtype]
    bads = [float('nan'), float('inf'), float('-inf')] + ([65504.0, -65504.0] if dtype == torch.float16 else [])
    emu = EmuBackend()
    for off, n in ((0, 1), (0, 7), (0, 8), (1, 9), (3, 4099), (0, 65536), (5, 69990)):
        x = base[off:off + n]                      # (off != 0: a pointer off the 16-byte grid - scalar head in front of the body)
        sc = torch.zeros(8, dtype=torch.float64, device=dev)
        x[n // 2] = big
        be.scaler_check(x, sc)
        assert float(sc[0]) == 0.0, (off, n, 'clean buffer flagged')
        for pos in sorted({0, n - 1, n // 3}):
            for bad in bads:
                keep = x[pos].clone()
                x[pos] = bad
                sc.zero_()
                be.scaler_check(x, sc)
                found = float(sc[0])
                ref = torch.zeros(8, dtype=torch.float64)
                emu.scaler_check(x.cpu(), ref)
                assert found > 0.0 and float(ref[0]) > 0.0, (off, n, pos, bad, found)
                x[pos] = keep
        x[n // 2] = 0.5


def test_scaler_step_and_the_identity_optimizer_step(be):
    """ase_hip_scaler_step: both outcomes against the emulator, then ase_hip_adam on what it wrote - a skipped step leaves weights
    and moments bit-for-bit alone, a clean one equals the plain optimizer step."""
    dev = 'cuda'
    emu = EmuBackend()
    g0 = torch.Generator().manual_seed(5)
    n = 10007
    w0 = torch.randn(n, generator=g0)
    m0, v0 = torch.randn(n, generator=g0) * 1e-3, torch.rand(n, generator=g0) * 1e-6
    gr = torch.randn(n, generator=g0) * 1e-2
    for found in (0.0, 3.0):
        opt = torch.tensor([4.0, 2e-5, 0.9, 0.999, 1e-8, 1 - 0.9 ** 4, 1 - 0.999 ** 4, 0.0], dtype=torch.float64)
        sc = torch.tensor([found, 2.0, 5.0, 9.0, 0, 0, 0, 0], dtype=torch.float64)
        eff = torch.zeros(8, dtype=torch.float64)
        d = lambda t: t.clone().to(dev)
        opt_d, sc_d, eff_d, g_d, w_d, m_d, v_d = d(opt), d(sc), d(eff), d(gr), d(w0), d(m0), d(v0)
        sc[4:8] = torch.tensor([4096.0, 2.0, 0.5, 6.0], dtype=torch.float64)        # scale, growth, backoff, interval (tracker at 5)
        sc_d = d(sc)
        tab_d, tab = torch.zeros(8, device=dev), torch.zeros(8)
        be.scaler_step(sc_d, opt_d, eff_d, g_d, scale_tab=tab_d)
        be.adam(w_d, g_d, m_d, v_d, eff_d)
        g_e, w_e, m_e, v_e = gr.clone(), w0.clone(), m0.clone(), v0.clone()
        emu.scaler_step(sc, opt, eff, g_e, scale_tab=tab)
        assert torch.equal(tab_d.cpu(), tab) and float(tab[0]) == (2048.0 if found else 8192.0)      # backoff | growth at the interval
        emu.adam(w_e, g_e, m_e, v_e, eff)
        torch.cuda.synchronize()
        assert torch.equal(sc_d.cpu(), sc) and torch.equal(opt_d.cpu(), opt) and torch.equal(eff_d.cpu(), eff)
        assert torch.equal(g_d.cpu(), g_e)
        if found:
            assert torch.equal(w_d.cpu(), w0) and torch.equal(m_d.cpu(), m0) and torch.equal(v_d.cpu(), v0)
            assert sc.tolist()[:5] == [0.0, 3.0, 0.0, 10.0, 2048.0] and float(opt[0]) == 3.0
        else:
            assert torch.allclose(w_d.cpu(), w_e, rtol=1e-6, atol=1e-9) and torch.allclose(m_d.cpu(), m_e, rtol=1e-6, atol=1e-12)
            assert torch.allclose(v_d.cpu(), v_e, rtol=1e-6, atol=1e-15) and not torch.equal(w_d.cpu(), w0)


def test_scaler_check_multi_and_fold(be):
    """ase_hip_scaler_check_multi: one launch over a table of buffers of mixed storage types, sizes and alignments finds a single bad
    element in any of them and nothing in clean ones; ase_hip_scaler_fold moves the records' counts into scaler[found]; a count alone
    makes ase_hip_scaler_step skip."""
    dev = 'cuda'
    g = torch.Generator().manual_seed(9)
    mk = lambda n, dt: (torch.rand(n, generator=g) * 2 - 1).to(dt).to(dev)
    pool16 = mk(300000, torch.float16)
    bufs = [mk(5, torch.float32), pool16[1:70001], mk(1 << 20, torch.bfloat16), pool16[80000:80009], mk(4099, torch.float32),
            mk(2 << 20, torch.float16)]
    table = be.make_check_table(bufs)
    sc = torch.zeros(8, dtype=torch.float64, device=dev)
    be.scaler_check_multi(bufs, sc, table=table)
    assert float(sc[0]) == 0.0
    for i, t in enumerate(bufs):
        for pos in (0, t.numel() - 1, t.numel() // 2):
            for bad in [float('nan'), float('-inf')] + ([65504.0] if t.dtype == torch.float16 else []):
                keep = t[pos].clone()
                t[pos] = bad
                sc.zero_()
                be.scaler_check_multi(bufs, sc, table=table)
                assert float(sc[0]) > 0.0, (i, pos, bad)
                t[pos] = keep
    sc.zero_()
    be.scaler_check_multi(bufs, sc, table=table)
    assert float(sc[0]) == 0.0
    tab = torch.tensor([8.0, 0.0, 0.125, 2.0, 1 / 64.0, 0.0, 1.0, 1.0], device=dev)
    be.scaler_fold(sc, tab)
    assert float(sc[0]) == 3.0 and tab.tolist() == [8.0, 0.0, 0.125, 0.0, 1 / 64.0, 0.0, 1.0, 0.0]
    # a producer's report alone (scaler[found] = 0) skips the step
    sc = torch.tensor([0.0, 0, 0, 0, 8.0, 2.0, 0.5, 100.0], dtype=torch.float64, device=dev)
    tab = torch.tensor([8.0, 0.0, 0.125, 0.0, 1 / 64.0, 1.0, 1.0, 0.0], device=dev)
    opt = torch.tensor([4.0, 2e-5, 0.9, 0.999, 1e-8, 0.3, 0.004, 0.0], dtype=torch.float64, device=dev)
    eff = torch.zeros(8, dtype=torch.float64, device=dev)
    gr = torch.ones(1000, device=dev)
    be.scaler_step(sc, opt, eff, gr, scale_tab=tab)
    assert float(gr.abs().sum()) == 0.0 and sc.tolist()[:5] == [0.0, 1.0, 0.0, 1.0, 4.0] and float(eff[1]) == 0.0
    assert tab.tolist() == [4.0, 0.0, 0.25, 0.0, 1 / 16.0, 0.0, 1.0, 0.0]


# (M, N, K): the phased 256 x 256 kernel | its 192-row form | the 4-wave kernel of long launches | 128 x 128 / 64 x 128 / 64 x 64 tiles with
# the row-per-lane epilogue | ragged rows through the LDS-slab epilogue | a narrow head
_REPORT_SHAPES = [(16384, 1024, 256), (12288, 1024, 128), (65536, 1024, 128), (4096, 512, 256), (4096, 1024, 128), (2048, 256, 128),
                  (300, 192, 256), (777, 64, 128)]


@pytest.mark.parametrize('dt', [torch.float16, torch.bfloat16])
@pytest.mark.parametrize('M,N,K', _REPORT_SHAPES)
def test_matrix_launches_report_what_they_store(be, dt, M, N, K):
    """ase_hip_gemm_nt with a scale record (ABI 7): ONE output that overflows the storage type - anywhere in the matrix, first / last row
    and column included - is reported into the record's count by whichever kernel the launch dispatches to; a clean product, the same
    product brought back into range by the record's factor, and an overflow a ReLU clips away are not; the emulator agrees on each."""
    from ase_amd import lib as L
    dev = 'cuda'
    g = torch.Generator().manual_seed(M + N + K)
    A0 = (torch.randn(M, K, generator=g) * 0.5)
    B0 = (torch.randn(N, K, generator=g) * 0.1)
    # planted row x column product big^2 K: f16 - beyond 65504 by itself; bf16 (whose largest finite value is f32's, and the accumulator
    # is f32) - finite, and the record's factor f_over carries it beyond; f_back: the factor that keeps it in range
    big, f_over, f_back = (40.0, 1.0, 2.0 ** -7) if dt == torch.float16 else (1.0e18, 4.0, 1.0)
    emu = EmuBackend()
    C = torch.zeros(M, N, dtype=dt, device=dev)
    bits = torch.zeros(M, N // 32, dtype=torch.int32, device=dev) if N % 32 == 0 else None

    def run(A, B, factor, act=L.ACT_NONE, mask=None):
        rec = torch.tensor([factor, 0.0], device=dev)
        be.gemm_nt(A.to(dt).to(dev), B.to(dt).to(dev), C, M, N, K, act=act, mask_out=mask, alpha_dev=rec)
        rec_e = torch.tensor([factor, 0.0])
        emu.gemm_nt(A.to(dt), B.to(dt), torch.zeros(M, N, dtype=dt), M, N, K, act=act, alpha_dev=rec_e)
        torch.cuda.synchronize()
        assert (float(rec[1]) > 0) == (float(rec_e[1]) > 0), (float(rec[1]), float(rec_e[1]))
        assert float(rec[0]) == factor
        return float(rec[1]) > 0
    assert not run(A0, B0, f_over)
    for r, c in ((0, 0), (M - 1, N - 1), (M // 2 + 1, N // 2 - 1), (M - 1, 0)):
        A, B = A0.clone(), B0.clone()
        A[r], B[c] = big, big
        assert run(A, B, f_over), (r, c)
        assert run(A, B, f_over, act=L.ACT_RELU, mask=bits), (r, c, 'relu')
        assert not run(A, B, -f_over, act=L.ACT_RELU, mask=bits), (r, c, 'clipped')       # the overflow is negative: ReLU stores 0
        assert not run(A, B, f_back), (r, c, 'scaled back')


def test_f32_storage_launches_and_loss_heads_report(be):
    """The 4-byte storage types (exact f32 and the three-product forms of the penalty's value path) report a non-finite output; the loss
    heads report a stored head gradient that saturates."""
    from ase_amd import lib as L
    dev = 'cuda'
    M, N, K = 4096, 512, 128
    g = torch.Generator().manual_seed(2)
    A = (torch.randn(M, K, generator=g) * 0.5).to(dev)
    B = (torch.randn(N, K, generator=g) * 0.1).to(dev)
    C = torch.zeros(M, N, device=dev)
    rec = torch.tensor([1.0, 0.0], device=dev)
    be.gemm_nt(A, B, C, M, N, K, alpha_dev=rec)
    assert float(rec[1]) == 0.0
    A[M - 1], B[N - 1] = 3.0e19, 3.0e19
    be.gemm_nt(A, B, C, M, N, K, alpha_dev=rec)
    assert float(rec[1]) > 0.0 and not bool(torch.isfinite(C[M - 1, N - 1]))
    # loss heads: the discriminator's stored logit gradient at an absurd scale
    logit = torch.zeros(12, 1, device=dev)
    d = torch.zeros(12, 1, dtype=torch.float16, device=dev)
    acc = torch.zeros(L.ACC_COUNT, dtype=torch.float64, device=dev)
    rec = torch.tensor([2.0 ** 40, 0.0], device=dev)
    be.disc_head(logit, d, None, acc, 4, 4, 5.0, grad_scale=1.0, dyn=rec)
    assert float(rec[1]) > 0.0 and float(d.float().abs().max()) == 65504.0
    rec = torch.tensor([2.0 ** 10, 0.0], device=dev)
    be.disc_head(logit, d, None, acc, 4, 4, 5.0, grad_scale=1.0, dyn=rec)
    assert float(rec[1]) == 0.0
    e = torch.randn(8, 64, generator=g).to(dev)
    z = torch.nn.functional.normalize(torch.randn(8, 64, generator=g), dim=-1).to(dev)
    de = torch.zeros(8, 64, dtype=torch.float16, device=dev)
    for factor, hit in ((2.0 ** 40, True), (1.0, False)):
        rec = torch.tensor([factor, 0.0], device=dev)
        be.enc_head(e, z, de, None, None, acc, 8, 8, 64, 5.0, grad_scale=1.0, dyn=rec)
        assert (float(rec[1]) > 0.0) == hit, factor


@pytest.mark.parametrize('name,gp_f32', [('ase_tiny', False), ('amp_tiny', False), ('ppo_tiny', False), ('ase_sep_gp_tiny', False),
                                         ('ase_tiny', True)])
def test_dynamic_loss_scale_on_gpu(be, name, gp_f32, golden_dir):
    G = torch.load(os.path.join(golden_dir, name + '.pt'), weights_only=False)
    eng = check_dynamic_loss_scale(G, lambda: be, device='cuda', gp_f32=gp_f32)
    assert eng.be is be


def test_mixed_precision_update_replays_programs_across_a_scale_change(golden_dir):
    """The agent under the reference's flag (mixed_precision: True) with recorded launch programs, from an overflowing initial scale:
    ONE step is skipped per backoff (here one backoff of 2^-36), the very next step of the same update runs at the new scale, and the
    recorded programs - which read the scale on the device - are replayed across the change without being dropped."""
    import copy
    from tests.test_agent_emu import replay_epochs
    from tests.test_scaler_emu import _agent_without_precision_key
    from ase_amd.backend import HipBackend
    G = torch.load(os.path.join(golden_dir, 'ase_tiny.pt'), weights_only=False)
    Gm = copy.deepcopy(G)
    Gm['cfg'].update(mixed_precision=True, graph_capture=True, loss_scaler={'init_scale': 2.0 ** 40, 'backoff_factor': 2.0 ** -36})
    ag = _agent_without_precision_key(Gm, device='cuda', backend=HipBackend())
    assert ag.engine.dyn_scale and ag.use_graph
    w0 = ag.model.a2c_network.flat_params.clone()
    Gm['epochs'] = Gm['epochs'] + copy.deepcopy(Gm['epochs'])               # four updates: (skip, then eager) | record | replay | replay
    dropped = []
    drop = ag._drop_graphs
    ag._drop_graphs = lambda: (dropped.append(1), drop())
    replay_epochs(Gm, ag, rtol=1.0, wtol=1.0, check=False)
    torch.cuda.synchronize()
    st = ag.engine.scaler_state()
    assert st['skipped'] == 1 and st['scale'] == 16.0, st                    # one step lost, not an update's worth
    assert float(ag.engine.opt_state[0]) == st['steps'] - 1
    assert not dropped
    w1 = ag.model.a2c_network.flat_params
    assert bool(torch.isfinite(w1).all()) and not torch.equal(w0, w1)
