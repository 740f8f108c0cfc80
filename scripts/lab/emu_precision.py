"""Lab: storage-precision feasibility on the CPU emulator (bf16 vs f16 vs f32) at full width, M = 2048.
Self-consistent old log-probabilities (the engine's own inference path in the same dtype)."""
import copy, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import restated as R
from tests.emu_backend import EmuBackend
from ase_amd.engine import UpdateEngine
from ase_amd.learning.network_builder import ASEBuilder
from ase_amd import cfg as defaults

M, AMB = int(os.environ.get('M', 2048)), int(os.environ.get('AMB', 512))
torch.set_num_threads(16)

def run(dt, selfc=True):
    net_p, cfg = defaults.get('ase')
    cfg = copy.deepcopy(cfg)
    cfg['minibatch_size'], cfg['amp_minibatch_size'] = M, AMB
    torch.manual_seed(0)
    b = ASEBuilder(); b.load(net_p)
    net = b.build('ase', actions_num=31, input_shape=(253,), num_seqs=1, value_size=1, amp_input_shape=(1400,),
                  ase_latent_shape=(64,), device='cpu')
    g = torch.Generator().manual_seed(1)
    z = torch.randn(M, 64, generator=g); nz = torch.randn(M, 64, generator=g)
    mb = {'obs': torch.randn(M, 253, generator=g) * 1.5 + 0.2, 'actions': torch.randn(M, 31, generator=g) * 0.1,
          'mu': torch.randn(M, 31, generator=g) * 0.05, 'sigma': torch.full((M, 31), 0.055023),
          'advantages': torch.randn(M, generator=g), 'old_values': torch.randn(M, 1, generator=g),
          'returns': torch.randn(M, 1, generator=g), 'rand_action_mask': (torch.rand(M, generator=g) < 0.8).float(),
          'ase_latents': z / z.norm(dim=-1, keepdim=True), 'amp_obs': torch.randn(M, 1400, generator=g),
          'amp_obs_replay': torch.randn(M, 1400, generator=g) * 1.1, 'amp_obs_demo': torch.randn(M, 1400, generator=g) + 0.3}
    nz = nz / nz.norm(dim=-1, keepdim=True)
    be = EmuBackend()
    eng = UpdateEngine('ase', net, cfg, be, minibatch=M, amp_minibatch=AMB, dtype=dt)
    sd = R.canonical_sd(net.state_dict(), False, requires_grad=[k for k, p in net.named_parameters() if p.requires_grad])
    sd = {k: v.cpu() if not v.requires_grad else v.detach().cpu().requires_grad_(True) for k, v in sd.items()}
    # rollout statistics: a large earlier batch of the same distribution (the update barely moves them)
    big = torch.randn(200000, 253, generator=g) * 1.5 + 0.2
    rms = {'obs': R.rms_update(R.rms_new(253), big), 'amp': R.rms_new(1400)}
    from tests.helpers import set_rms
    set_rms(eng.obs_state, rms['obs'])
    with torch.no_grad():
        if selfc:
            mu0 = eng.policy_forward(mb['obs'], mb['ase_latents'], want=('mu',))['mu'].clone()
        else:
            o = R.rms_normalize(rms['obs'], mb['obs'])
            mu0, _ = R.eval_actor('ase', sd, o, mb['ase_latents'])
        ls0 = torch.full((31,), -2.9)
        mb['actions'] = mu0 + torch.exp(ls0) * torch.randn(M, 31, generator=g)
        mb['mu'] = mu0.clone()
        mb['old_logp_actions'] = R.neglogp(mb['actions'], mu0, torch.exp(ls0), ls0)
    sd64 = {k: (v.detach().double().requires_grad_(True) if v.requires_grad else v.double()) for k, v in sd.items()}
    rms64 = {'obs': R.rms_clone(rms['obs']), 'amp': R.rms_new(1400)}
    ref64 = R.calc_gradients('ase', sd64, rms64, {k: v.double() for k, v in mb.items()}, cfg, nz.double())
    idx = torch.arange(M, dtype=torch.int32)
    streams = [(mb['amp_obs'], idx, (0, 0)), (mb['amp_obs_replay'], idx, (0, 0)), (mb['amp_obs_demo'], idx, (0, 0))]
    eng.step(mb, idx, (0, 0), streams, new_z=nz, apply=False)
    res, grads = eng.results(), eng.export_grads()
    scale = {'actor_loss': 1.0, 'enc_loss': 1.0}
    out = {}
    for k in ('actor_loss', 'critic_loss', 'b_loss', 'disc_loss', 'disc_grad_penalty', 'enc_loss', 'amp_diversity_loss',
              'kl', 'entropy', 'actor_clip_frac', 'disc_agent_acc', 'disc_demo_acc'):
        sc = max(abs(float(ref64[k])), scale.get(k, 0.0))
        out[k] = abs(float(res[k]) - float(ref64[k])) / (sc + 1e-30)
    rels = {}
    for k, p in sd64.items():
        if p.requires_grad:
            rels[k] = float((grads[k].double() - p.grad).norm() / p.grad.norm())
    return out, rels, {k: float(ref64[k]) for k in out}

for name, dt in (('bf16', torch.bfloat16), ('f16', torch.float16), ('f32', torch.float32)):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    o, r, ref = run(dt)
    print(name, 'loss rel:', {k: f'{v:.1e}' for k, v in o.items()})
    print(name, 'worst loss', max(o.values()), 'grad rel l2 worst', max(r.values()), max(r, key=r.get),
          'median', sorted(r.values())[len(r) // 2])
    if name == 'bf16':
        print('ref', ref)
