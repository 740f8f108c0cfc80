"""GPU tests written after round 3's GPU budget was spent: their first hardware run is the driver's.  They sort last (a
fault here cannot disturb the validated suite) and are xfail(strict=False): a pass is reported as XPASS, a failure as XFAIL -
neither turns the suite red, both are visible in the report.  Promote them once seen green."""
import os

import pytest
import torch

from ase_amd import lib as L
from tests.emu_backend import EmuBackend
from tests.helpers import close
from tests.test_agent_emu import make_agent, replay_epochs

pytestmark = pytest.mark.gpu
FIRST_RUN = pytest.mark.xfail(strict=False, reason="first hardware run (written without GPU access at the end of round 3)")


@pytest.fixture(scope='module')
def be():
    from ase_amd.backend import HipBackend
    return HipBackend()


@FIRST_RUN
@pytest.mark.parametrize('kl,expect', [(0.05, 2e-5 / 1.5), (0.001, 2e-5 * 1.5), (0.01, 2e-5)])
def test_finalize_scalars_adaptive_lr(be, kl, expect):
    """The rl_games AdaptiveScheduler branch of ase_hip_finalize_scalars (learning/common_agent.py:204-208): lr / 1.5 when the
    step's kl exceeds 2 x kl_threshold, x 1.5 below half of it, unchanged between - device result = emulator = closed form."""
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


@FIRST_RUN
@pytest.mark.parametrize('act', ['elu', 'gelu', 'softplus', 'selu', 'sigmoid'])
def test_activation_family_f32(be, act, golden_dir):
    """The reference agent with `activation: <act>` in every MLP (oracle/make_golden.py acts; emulator twin:
    tests/test_agent_emu.py::test_activation_family_against_the_reference): a whole update on the GPU in f32, incl. the
    gradient penalty's double backward through the curved activation."""
    G = torch.load(os.path.join(golden_dir, f'ase_{act}_tiny.pt'), weights_only=False)
    ag = make_agent(G, be, device='cuda', precision='f32')
    replay_epochs(G, ag, rtol=3e-4, wtol=float(G['cfg']['learning_rate']) * 0.25)
