"""Data-parallel update (minibatch rows sharded over ranks, SUM all-reduce of the flat gradient buffer with global
denominators, all-reduced normaliser moments / mask sums): world_size-2 gloo run on CPU (op emulator) must reproduce
the single-rank result, which itself matches the reference (tests/test_agent_emu.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.emu_backend import EmuBackend
from tests.test_agent_emu import make_agent, replay_epochs

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    ag = make_agent(G, EmuBackend(), world_size=world, rank=rank)
    infos = replay_epochs(G, ag, rtol=3e-4, wtol=G['cfg']['learning_rate'] * 0.25, check=True)   # vs the REFERENCE
    if rank == 0:
        torch.save({'flat': ag.model.a2c_network.flat_params.clone(), 'kl': torch.stack([x.float() for x in infos[-1]['kl']])},
                   out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('name', ['ase_tiny', 'amp_tiny', 'ppo_tiny', 'ase_gp_tiny'])
def test_two_ranks_match_reference_and_single_rank(name, tmp_path):
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    ag1 = make_agent(G, EmuBackend())
    replay_epochs(G, ag1, rtol=2e-4, wtol=G['cfg']['learning_rate'] * 0.05)
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(2, _free_port(), name, out), nprocs=2, join=True)
    r = torch.load(out)
    # SUM-of-partials vs one pass: f32 summation order differs, compare at 1e-5 of the weight scale
    assert torch.allclose(r['flat'], ag1.model.a2c_network.flat_params, rtol=1e-5, atol=G['cfg']['learning_rate'] * 0.25)


# ------------------------------------------------------------------------------------------------ round 2
def _free_run(ag, G, epochs=2):
    """Two updates with EVERYTHING drawn by the agent itself (dataset permutations, ring sample permutations, replay keep
    masks, diversity latents on the 'device'): only the experience and the demo stream are given."""
    from tests.test_agent_emu import regenerate
    regenerate(G)
    kind = G['kind']
    if kind != 'ppo':
        ag.vec_env.q.append(G['demo_init'].clone())
    infos = []
    for E in G['epochs'][:epochs]:
        ag.update_epoch()
        for k, v in E['exp'].items():
            if k in ag.experience:
                ag.experience[k].copy_(v)
        batch = ag._play_steps_tail()
        if kind != 'ppo':
            ag.vec_env.q.append(E['demo_fetched'].clone())
        infos.append(ag.update(batch))
    return infos


def _worker_free(rank, world, port, name, out, mode):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    torch.manual_seed(1000 + 17 * rank)                  # the ranks' global torch RNG streams DIFFER on purpose
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    ag = make_agent(G, EmuBackend(), world_size=world, rank=rank, seed=5, dp_mode=mode)
    if rank == 1:
        # a rank that starts (or restores) with different weights / statistics is overwritten by rank 0's
        # (rl_games HorovodWrapper.setup_algo: broadcast_parameters + broadcast_optimizer_state)
        ag.model.a2c_network.flat_params.add_(0.01)
        ag.engine.obs_state[0] = 3.0
    ag._sync_initial_state()                             # (a collective: every rank calls it, as in train() / restore())
    infos = _free_run(ag, G)
    if mode == 'horovod':
        ag.curr_frames = ag.batch_size
        ag.sync_stats()
    flat = ag.model.a2c_network.flat_params.clone()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    if rank == 0:
        torch.save({'flat': flat, 'others': gathered, 'kl': torch.stack([x.float() for x in infos[-1]['kl']]),
                    'obs_state': ag.engine.obs_state.clone(), 'frames': ag.curr_frames}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('name', ['ase_tiny', 'amp_tiny'])
def test_two_ranks_draw_like_one_rank(name, tmp_path):
    """Sharded mode with NOTHING injected: dataset permutations, ring sample permutations, keep masks come from the agents'
    shared-seed generator, the diversity latents from the counter-based stream indexed by the GLOBAL row - so two ranks
    (whose global torch RNGs differ) reproduce the one-rank update, and rank 1's deliberately wrong initial weights are
    replaced by rank 0's."""
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    torch.manual_seed(4242)
    ag1 = make_agent(G, EmuBackend(), seed=5)
    _free_run(ag1, G)
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker_free, args=(2, _free_port(), name, out, 'shard'), nprocs=2, join=True)
    r = torch.load(out)
    lr = G['cfg']['learning_rate']
    assert torch.equal(r['others'][0], r['others'][1])                     # replicas identical
    assert torch.allclose(r['flat'], ag1.model.a2c_network.flat_params, rtol=1e-5, atol=lr * 0.25)
    assert torch.allclose(r['obs_state'], ag1.engine.obs_state, rtol=1e-9, atol=1e-9)


@pytest.mark.timeout(300)
def test_horovod_mode_semantics(tmp_path):
    """The reference's own multi-GPU semantics (rl_games HorovodWrapper): every rank updates on its OWN minibatches with
    local statistics, gradients are averaged (replicas stay identical), running statistics are averaged per epoch and the
    frame counters summed (learning/common_agent.py:94-107)."""
    name = 'amp_tiny'
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker_free, args=(2, _free_port(), name, out, 'horovod'), nprocs=2, join=True)
    r = torch.load(out)
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    assert torch.equal(r['others'][0], r['others'][1])                     # averaged gradients: identical replicas
    assert bool(torch.isfinite(r['flat']).all()) and bool(torch.isfinite(r['kl']).all())
    H, N = G['cfg']['horizon_length'], G['spec']['num_envs']
    assert r['frames'] == 2 * H * N                                        # summed over the two ranks


# ------------------------------------------------------------------------------------------------ round 3
def _worker_gp32(rank, world, port, name, out):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    ag = make_agent(G, EmuBackend(), precision='f16gp32', world_size=world, rank=rank)
    infos = replay_epochs(G, ag, rtol=0, wtol=0, check=False)
    if rank == 0:
        torch.save({'flat': ag.model.a2c_network.flat_params.clone(),
                    'gp0': float(infos[0]['disc_grad_penalty'][0])}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_f16gp32_penalty_is_the_references(tmp_path):
    """The f32 value path of the gradient penalty under row sharding: rank r runs it on its AMB / R demo rows, the squared
    norms are summed with the other loss partial sums - the reported penalty of the first step is the reference's, and the
    two-rank run ends where the one-rank run of the same precision ends."""
    name = 'ase_tiny'
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    ag1 = make_agent(G, EmuBackend(), precision='f16gp32')
    replay_epochs(G, ag1, rtol=0, wtol=0, check=False)
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker_gp32, args=(2, _free_port(), name, out), nprocs=2, join=True)
    r = torch.load(out)
    ref = float(G['epochs'][0]['steps'][0]['disc_grad_penalty'])
    assert abs(r['gp0'] - ref) <= 1e-5 * abs(ref), (r['gp0'], ref)
    assert torch.allclose(r['flat'], ag1.model.a2c_network.flat_params, rtol=1e-4, atol=G['cfg']['learning_rate'] * 1.0)


# ------------------------------------------------------------------------------------------------ round 4
def _worker_adaptive(rank, world, port, name, out):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    torch.manual_seed(1000 + 17 * rank)
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    # a threshold the ranks' LOCAL kls straddle differently: rank-local schedules would diverge within a few steps
    ag = make_agent(G, EmuBackend(), world_size=world, rank=rank, seed=5, dp_mode='horovod', lr_schedule='adaptive',
                    kl_threshold=2e-4)
    ag._sync_initial_state()
    infos = _free_run(ag, G)
    lr = ag.engine.opt_state[1:2].clone()
    lrs = [torch.zeros_like(lr) for _ in range(world)]
    dist.all_gather(lrs, lr)
    kl = torch.stack([x.float().reshape(()) for x in infos[-1]['kl']])
    kls = [torch.zeros_like(kl) for _ in range(world)]
    dist.all_gather(kls, kl)
    flat = ag.model.a2c_network.flat_params.clone()
    flats = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(flats, flat)
    if rank == 0:
        torch.save({'lrs': lrs, 'kls': kls, 'flats': flats, 'lr0': G['cfg']['learning_rate'],
                    'last_lr': [float(x) for x in infos[-1]['last_lr']]}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_horovod_adaptive_lr_is_the_same_on_every_rank(tmp_path):
    """learning/amp_agent.py:224-228: under the legacy schedule the step's kl is averaged over the ranks BEFORE the scheduler
    sees it - every rank derives the same learning rate and the replicas stay identical (round 3's advisor found each rank
    scheduling from its local kl)."""
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker_adaptive, args=(2, _free_port(), 'amp_tiny', out), nprocs=2, join=True)
    r = torch.load(out)
    assert torch.equal(r['lrs'][0], r['lrs'][1])                          # one schedule
    assert torch.equal(r['kls'][0], r['kls'][1])                          # the reported kl is the rank average
    assert torch.equal(r['flats'][0], r['flats'][1])                      # replicas identical
    assert float(r['lrs'][0]) != r['lr0']                                 # (the schedule did move in this run)


# ------------------------------------------------------------------------------------------------ dynamic loss scale
def _worker_scaler(rank, world, port, name, out):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    be = EmuBackend()
    ag = make_agent(G, be, precision='f16', world_size=world, rank=rank, loss_scale='dynamic', loss_scaler={'init_scale': 16.0})
    assert ag.engine.dyn_scale and ag.engine.gs == 1.0 and ag.engine.scaler_state()['scale'] == 16.0
    if rank == 1:            # an overflow only THIS rank sees, in the very first step
        orig, n = be.scaler_check, [0]

        def check(buf, scaler):
            if n[0] == 0:
                scaler[0] += 1.0
            n[0] += 1
            orig(buf, scaler)
        be.scaler_check = check
    replay_epochs(G, ag, rtol=1.0, wtol=1.0, check=False, max_steps=2)
    flat = ag.model.a2c_network.flat_params.clone()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    st = ag.engine.scaler_state()
    states = [None] * world
    dist.all_gather_object(states, st)
    if rank == 0:
        torch.save({'flats': gathered, 'states': states, 'opt_step': float(ag.engine.opt_state[0])}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_skip_the_same_step(tmp_path):
    """loss_scale: dynamic under data parallelism: the overflow flag is exchanged (SUM) before the decision, so a step one rank
    found an overflow in is skipped by BOTH, the weights stay identical across the ranks and the scale moves on both."""
    out = str(tmp_path / 'sc.pt')
    mp.spawn(_worker_scaler, args=(2, _free_port(), 'ase_tiny', out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    assert torch.equal(r['flats'][0], r['flats'][1])
    s0, s1 = r['states']
    assert s0 == s1, (s0, s1)
    G = torch.load(os.path.join(GOLDEN, 'ase_tiny.pt'), weights_only=False)
    n_upd = len(G['epochs'])
    assert s0['steps'] == 2 * n_upd and s0['skipped'] == 1 and s0['scale'] == 8.0       # one backoff, right behind the skipped step
    assert r['opt_step'] == 2 * n_upd - 1                                                # the skipped step was no optimizer step


# ------------------------------------------------------------------------------------------------ round 6: the exchange itself
def _worker_exchange(rank, world, port, name, out, grad_dtype):
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    G = torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)
    ag = make_agent(G, EmuBackend(), world_size=world, rank=rank, dp_grad_dtype=grad_dtype)
    calls = []
    orig = dist.all_reduce

    def counted(t, *a, **kw):
        calls.append(t.numel() * t.element_size())
        return orig(t, *a, **kw)
    dist.all_reduce = counted
    tol = dict(rtol=3e-4, wtol=G['cfg']['learning_rate'] * 0.25) if grad_dtype == 'f32' else dict(rtol=5e-2, wtol=G['cfg']['learning_rate'] * 2.2)
    infos = replay_epochs(G, ag, check=grad_dtype == 'f32', **tol)
    dist.all_reduce = orig
    steps = sum(len(i['kl']) for i in infos)
    if rank == 0:
        torch.save({'flat': ag.model.a2c_network.flat_params.clone(), 'calls': calls, 'steps': steps,
                    'bucket_bytes': 4 * ag.engine.n_train}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_three_collectives_per_step_and_the_bf16_payload(tmp_path):
    """The sharded step's exchange: ONE statistics collective [amp sums | obs sums | mask sum], the discriminator's gradient
    bucket, the policy's gradient bucket carrying the loss partial sums as (hi, lo) f32 pairs - three collectives per optimisation
    step (round 5: six) and still the reference's result; dp_grad_dtype = 'bf16' sends half the gradient bytes and stays within
    Adam's +-lr of the f32 exchange."""
    G = torch.load(os.path.join(GOLDEN, 'ase_tiny.pt'), weights_only=False)
    res = {}
    for dt in ('f32', 'bf16'):
        out = str(tmp_path / (dt + '.pt'))
        mp.spawn(_worker_exchange, args=(2, _free_port(), 'ase_tiny', out, dt), nprocs=2, join=True)
        res[dt] = torch.load(out, weights_only=False)
    f, b = res['f32'], res['bf16']
    per_epoch_extra = 4          # once per epoch: the sharded reward inference + statistics of the tail (not per step)
    assert len(f['calls']) <= 3 * f['steps'] + per_epoch_extra * len(G['epochs']), (len(f['calls']), f['steps'])
    assert len(b['calls']) <= 4 * b['steps'] + per_epoch_extra * len(G['epochs'])        # (bf16 payload: the f64 partial sums travel apart)
    grad_f = sum(x for x in f['calls'] if x > 0.2 * f['bucket_bytes'])
    grad_b = sum(x for x in b['calls'] if x > 0.1 * b['bucket_bytes'])
    assert 0.45 * grad_f < grad_b < 0.55 * grad_f, (grad_f, grad_b)                         # half the bytes on the links
    lr = G['cfg']['learning_rate']
    d = (f['flat'] - b['flat']).abs()
    assert float(d.max()) <= 2.2 * lr * len(G['epochs']) * 8 and float((d > 0.5 * lr).float().mean()) < 0.2
