import copy, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import restated as R
from ase_amd.engine import UpdateEngine
from ase_amd.backend import HipBackend
from ase_amd.learning.network_builder import ASEBuilder
from tests.test_gpu_engine import _ase_full_cfg
be = HipBackend()
M, AMB = 2048, 512
net_p, cfg = _ase_full_cfg(); cfg = copy.deepcopy(cfg); cfg['minibatch_size'], cfg['amp_minibatch_size'] = M, AMB
for dt in (torch.float32, torch.bfloat16):
    torch.manual_seed(0)
    b = ASEBuilder(); b.load(net_p)
    net = b.build('ase', actions_num=31, input_shape=(253,), num_seqs=1, value_size=1, amp_input_shape=(1400,), ase_latent_shape=(64,), device='cuda')
    g = torch.Generator().manual_seed(1)
    z = torch.randn(M, 64, generator=g); nz = torch.randn(M, 64, generator=g)
    mb = {'obs': torch.randn(M, 253, generator=g) * 1.5 + 0.2, 'actions': torch.randn(M, 31, generator=g) * 0.1,
          'mu': torch.randn(M, 31, generator=g) * 0.05, 'sigma': torch.full((M, 31), 0.055023),
          'advantages': torch.randn(M, generator=g), 'old_values': torch.randn(M, 1, generator=g),
          'returns': torch.randn(M, 1, generator=g), 'rand_action_mask': (torch.rand(M, generator=g) < 0.8).float(),
          'ase_latents': z / z.norm(dim=-1, keepdim=True), 'amp_obs': torch.randn(M, 1400, generator=g),
          'amp_obs_replay': torch.randn(M, 1400, generator=g) * 1.1, 'amp_obs_demo': torch.randn(M, 1400, generator=g) + 0.3}
    nz = nz / nz.norm(dim=-1, keepdim=True)
    sd = R.canonical_sd(net.state_dict(), False, requires_grad=[k for k, p in net.named_parameters() if p.requires_grad])
    sd = {k: v.cpu() if not v.requires_grad else v.detach().cpu().requires_grad_(True) for k, v in sd.items()}
    rms = {'obs': R.rms_new(253), 'amp': R.rms_new(1400)}
    with torch.no_grad():
        o = R.rms_normalize(R.rms_update(R.rms_clone(rms['obs']), mb['obs']), mb['obs'])
        mu0, ls0 = R.eval_actor('ase', sd, o, mb['ase_latents'])
        mb['actions'] = mu0 + torch.exp(ls0) * torch.randn(M, 31, generator=g)
        mb['mu'] = mu0 + 0.01 * torch.randn(M, 31, generator=g)
        mb['old_logp_actions'] = R.neglogp(mb['actions'], mu0, torch.exp(ls0), ls0) + 0.1 * torch.randn(M, generator=g)
    ref = R.calc_gradients('ase', sd, rms, mb, cfg, nz)
    sd64 = {k: (v.detach().double().requires_grad_(True) if v.requires_grad else v.double()) for k, v in sd.items()}
    mb64 = {k: v.double() for k, v in mb.items()}
    ref64 = R.calc_gradients('ase', sd64, {'obs': R.rms_new(253), 'amp': R.rms_new(1400)}, mb64, cfg, nz.double())
    eng = UpdateEngine('ase', net, cfg, be, minibatch=M, amp_minibatch=AMB, dtype=dt)
    idx = torch.arange(M, dtype=torch.int32, device='cuda')
    mbg = {k: v.cuda() for k, v in mb.items()}
    streams = [(mbg['amp_obs'], idx, (0, 0)), (mbg['amp_obs_replay'], idx, (0, 0)), (mbg['amp_obs_demo'], idx, (0, 0))]
    eng.step(mbg, idx, (0, 0), streams, new_z=nz.cuda(), apply=False)
    torch.cuda.synchronize()
    res, grads = eng.results(), eng.export_grads()
    print('=====', dt)
    for k in ('actor_loss', 'critic_loss', 'b_loss', 'disc_loss', 'disc_grad_penalty', 'enc_loss', 'amp_diversity_loss', 'kl', 'entropy', 'actor_clip_frac', 'disc_agent_acc', 'disc_demo_acc'):
        print(f'{k:22s} hip {float(res[k]): .7e} cpu32 {float(ref[k]): .7e} cpu64 {float(ref64[k]): .7e}')
    for k, p in sd.items():
        if p.requires_grad:
            g32, g64, gh = p.grad.double(), sd64[k].grad, grads[k].cpu().double()
            mx = float(g64.abs().max())
            print(f'{k:36s} max {mx:.3e}  |hip-cpu64|/max {float((gh-g64).abs().max())/mx:.2e}  |cpu32-cpu64|/max {float((g32-g64).abs().max())/mx:.2e}  rms-rel hip {float((gh-g64).norm()/g64.norm()):.2e} cpu32 {float((g32-g64).norm()/g64.norm()):.2e}')
