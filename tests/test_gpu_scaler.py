"""GPU tests of the dynamic loss scale (csrc/scaler.hip behind config loss_scale: 'dynamic' - torch.cuda.amp.GradScaler's
found_inf / skipped step of the reference's mixed_precision path, learning/ase_agent.py:271-288): the two entry points against
the emulator's semantics (incl. the per-step backoff / growth of the device-resident scale, ABI 6), and the engine's skip -> backoff
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
        tab_d, tab = torch.zeros(4, device=dev), torch.zeros(4)
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
