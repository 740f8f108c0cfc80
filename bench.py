"""Benchmark of the ASE PPO-update hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--precision auto|f16gpx3|f16gp32|f32|bf16|...] [--no-graph]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): ASE PPO-update samples/sec, 4096 envs x horizon 32 (config 2 = the reference's
ase_humanoid.yaml verbatim: [1024,1024,512] actor/critic/disc MLPs, latent 64, obs 253, amp obs 1400,
minibatch 16384, amp minibatch 4096, 6 mini-epochs => 48 optimisation steps).

One "step" = one whole update of one rollout batch = everything ``train_epoch`` does after the simulator
loop: AMP/encoder rewards, GAE, advantage + value normalisation, demo/replay sampling, 48 minibatch
optimisation steps (normalisers, 4 MLPs forward/backward, losses incl. gradient penalty + diversity, Adam),
replay store.  Inputs (the synthetic experience buffer) are resident in HBM before the timed
region.  With N > 1 (one process per GPU) the default is the reference's own multi-GPU semantics (rl_games HorovodWrapper,
learning/common_agent.py:94-107): every rank owns 4096 environments and draws its own 16384-row minibatches, the gradient
buckets are averaged by RCCL all-reduces over xGMI (the discriminator bucket while the policy branch is still in its
backward) - weak scaling, `value` = all ranks' samples / max-over-ranks time.  `--dp-mode shard` row-shards every minibatch
of the SAME 4096 environments over the ranks instead (the R-rank update equals the 1-rank update; strong scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ase_amd          # noqa: E402
ase_amd.configure()     # the runtime's hardware-queue count, before HIP initialises (ase_amd/__init__.py); reported on the JSON line
import torch            # noqa: E402

MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'f16': 2500.0, 'f16gp32': 2500.0, 'f16gpx3': 2500.0, 'f32': 157.3, 'bf16x3': 2500.0 / 3}   # x3: three bf16 MFMAs per product        # /opt/skills/guides/MI355X_MICROARCH.md (dense)


# what the MFMA path multiplies in, per precision mode (the JSON line's `dtype`), and the mode in one sentence
DTYPE_OF = {'bf16': 'bf16', 'f16': 'f16', 'f16gp32': 'f16', 'f16gpx3': 'f16', 'f32': 'f32', 'bf16x3': 'bf16'}
MODE_NOTE = {
    'bf16': 'bf16 storage / MFMA, f32 accumulate, f32 heads and losses',
    'f16': 'IEEE-half storage / MFMA, f32 accumulate, f32 heads and losses, static gradient scale (= the reference\'s mixed_precision)',
    'f16gp32': 'f16 engine; the gradient penalty\'s value path (demo-row forward, chain) in exact f32',
    'f16gpx3': 'f16 engine (half storage / MFMA, f32 accumulate, static gradient scale); the gradient penalty\'s value path '
               '(6 launches of 4096 rows) as three f16 MFMAs per product on hi/lo half splits of power-of-two scaled f32 operands (ASE_F32H3)',
    'f32': 'f32 storage, exact-f32 MFMA', 'bf16x3': 'f32 storage, three bf16 MFMAs per product'}
PARITY_TOL = 1e-4          # BASELINE.json north_star: "losses matching the reference CPU path to rtol 1e-4"
COUNT_TOL = 1e-3           # the three counting statistics move in steps of 1 / rows: absolute (fresh rollout)
# ... and 5e-3 in the stress state: there 90 % of the samples sit beyond the PPO clip boundary |ratio - 1| > e_clip and the ratios are
# e^(O(10)) wide, so a 1e-4 relative error of a ratio moves a few dozen of the 16384 samples across the threshold (measured over
# this round's runs: 7e-5 ... 2.7e-3 for the half modes, 1.4e-4 for the f32 engine against the f32 oracle)
COUNT_TOL_STATE = {'fresh': COUNT_TOL, 'stress': 5e-3}
# The bar as it is APPLIED to a loss scalar x against the oracle's r:  |x - r| <= PARITY_TOL * max(|r|, floor)  - i.e. rtol 1e-4 with
# an absolute tolerance atol = 1e-4 * floor underneath.  floor = 1 for the two scalars that are means of signed O(1) summands
# whose VALUE is a small remainder (actor_loss = mean(-A r): normalised advantages have zero mean; enc_loss = -mean <z, e>), 0 for
# the others.  It is an atol, and the line says so (`parity.tol`); the TRUE relative error |x - r| / |r| of every scalar is
# reported beside it (`max_loss_true_rel`), and `strict_mode` is the mode that needs no floor at all.
PARITY_ATOL = {'actor_loss': 1e-4, 'enc_loss': 1e-4}
# The bar applies to the TERMS OF THE LOSS (learning/ase_agent.py:228-258: actor, critic, bound, entropy, discriminator incl. gradient
# penalty and logit regulariser, encoder, diversity).  `kl` is reported by the reference beside them but enters no loss and no
# gradient: it drives the adaptive schedule through factor-2 thresholds (kl > 2 x / < 0.5 x kl_threshold).  It is held to the same
# 1e-4 on a fresh rollout; in the stress state - 96 optimisation steps on one rollout, kl ~ 1 = 125 x the schedule's threshold, a
# state PPO training never visits - the half-storage engines carry 1e-4 ... 3e-4 on it: kl there is <(mu_new - mu_old)^2> / 2 sigma^2
# with sigma = e^-2.9, and the 2^-12 rounding of the weight shadows is a SYSTEMATIC error of mu_new (the same for every row) that
# does not average out over the minibatch (scripts/lab/grad_error_sources.py / DESIGN 3.2: weights alone 2.3e-4, activations
# 8e-5).  Its stress tolerance is therefore 1e-3, stated here and in the line (`parity.tol.kl`), never folded into the loss bar.
LOSS_TERMS = ('actor_loss', 'critic_loss', 'b_loss', 'entropy', 'disc_loss', 'disc_grad_penalty', 'disc_logit_loss', 'enc_loss',
              'amp_diversity_loss')
KL_TOL = {'fresh': 1e-4, 'stress': 1e-3}


def _loss_terms_rel(p):
    """(worst relative error over the terms of the loss, its name) - `kl` and the counting statistics are judged apart."""
    lr = p['loss_rel']
    k = max((k for k in LOSS_TERMS if k in lr), key=lambda k: lr[k])
    return lr[k], k


def _state_ok(p, state=None):
    """One rollout state of one mode against the bar: every term of the loss within max(rtol |ref|, atol), kl within its stated
    tolerance for that state, the three counting statistics within 1e-3 absolute."""
    state = state or ('stress' if 'stale' in (p.get('state') or '') else 'fresh')
    return _loss_terms_rel(p)[0] <= PARITY_TOL and p['loss_rel'].get('kl', 0.0) <= KL_TOL[state] and p['max_count_stat_abs'] <= COUNT_TOL_STATE[state]


def qualifying_mode(modes):
    """The fastest measured mode whose ten continuous loss scalars are all within the bar (PARITY_TOL / PARITY_ATOL) of the f32
    reference arithmetic on the first step of a FRESH rollout AND in the stale-rollout STRESS state, counting statistics within
    1e-3 absolute - in this run.  (Round 4 qualified on the fresh state alone and reported `stress_ok` beside it.)"""
    ok = [m for m, r in modes.items() if r.get('parity') and _state_ok(r['parity']['fresh'], 'fresh') and _state_ok(r['parity']['stress'], 'stress')]
    if not ok:
        return None
    q = max(ok, key=lambda m: modes[m]['value'])
    r = modes[q]
    f, st = r['parity']['fresh'], r['parity']['stress']
    out = {'precision': q, 'value': r['value'], 'unit': 'samples/s', 'ms_per_step': r['ms_per_step'],
           'criterion': 'all 10 continuous loss scalars within max(1e-4 |ref|, atol) of the reference arithmetic (f32 CPU oracle) on the '
                        'first step of a fresh rollout AND in the stress state; the 3 counting statistics within 1e-3 absolute',
           'fresh_max_loss_rel': f['max_loss_rel'], 'fresh_max_loss_rel_without_grad_penalty': f['max_loss_rel_without_grad_penalty'],
           'fresh_trajectory_max_loss_rel': f['trajectory']['max_loss_rel'], 'stress_max_loss_rel': st['max_loss_rel'],
           'stress_ok': _state_ok(st, 'stress'), 'fresh_worst_grad_rel_l2': f['worst_grad_rel_l2'],
           'by_mode': {m: {'fresh_max_loss_rel': x['parity']['fresh']['max_loss_rel'],
                           'worst_scalar': x['parity']['fresh']['max_loss_rel_scalar'],
                           'without_grad_penalty': x['parity']['fresh']['max_loss_rel_without_grad_penalty'],
                           'stress_max_loss_rel': x['parity']['stress']['max_loss_rel'], 'value': x['value']}
                       for m, x in modes.items() if x.get('parity')}}
    return out


def write_detail(full, path):
    """Everything the run measured (per-mode parity tables, per-kind GEMM classes, HBM kernels, trajectories) goes to a side
    file and to stderr; the stdout line stays small."""
    txt = json.dumps(full)
    print('[bench] detail: ' + txt, file=sys.stderr, flush=True)
    if not path:
        return None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        with open(path, 'w') as f:
            f.write(txt + '\n')
        return os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT) else path
    except OSError as e:
        print(f'[bench] could not write {path}: {e}', file=sys.stderr)
        return None


def _parity_scalars(p):
    """One parity state as a handful of scalars (the per-scalar tables stay in the detail file)."""
    if not p:
        return None
    lt, ltk = _loss_terms_rel(p)
    return {'max_loss_term_rel': lt, 'loss_term': ltk, 'kl_rel': p['loss_rel'].get('kl'),
            'max_loss_rel': p['max_loss_rel'], 'scalar': p['max_loss_rel_scalar'],
            'max_loss_true_rel': p.get('max_loss_true_rel'), 'true_rel_scalar': p.get('max_loss_true_rel_scalar'),
            'max_count_stat_abs': p['max_count_stat_abs'], 'worst_grad_rel_l2': p['worst_grad_rel_l2'],
            'median_grad_rel_l2': p['median_grad_rel_l2'],
            'grad_at_engine_masks': {k: (p.get('grad_at_engine_masks') or {}).get(k) for k in ('median_grad_rel_l2', 'worst_grad_rel_l2',
                                                                                              'flipped_mask_fraction')}
            if p.get('grad_at_engine_masks') else None,
            'trajectory_max_loss_rel': (p.get('trajectory') or {}).get('max_loss_rel'),
            'trajectory_steps': (p.get('trajectory') or {}).get('steps'), 'ok': _state_ok(p)}


def compact_line(full, detail_path=None):
    """The ONE stdout line of the contract: scalars only, < 5 KB (the driver keeps an 8 KB tail of stdout; round 3's 21 KB
    line was cut and could not be parsed).  tests/test_scripts.py holds the size bound on a canned result."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
            'dtype', 'data', 'config')
    out = {k: full[k] for k in keep}
    out['config'] = {k: v for k, v in full['config'].items()}
    r = full.get('roofline')
    if r:
        out['roofline'] = {k: r.get(k) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source',
                                                 'launches', 'avg_launch_us', 'algorithmic_flop_per_launch', 'algorithmic_bytes_per_launch',
                                                 'sustained_clock_mhz', 'frac_at_sustained_clock', 'share_of_gemm_time')}
        ag = r.get('all_gemm') or {}
        out['roofline']['all_gemm_frac'] = ag.get('frac')
        out['roofline']['gemm_launches_per_update'] = ag.get('launches_per_update')
        out['roofline']['algorithmic_tflop_per_update'] = ag.get('algorithmic_tflop_per_update')
        if r.get('main_loop'):
            out['roofline']['main_loop_mfma_busy'] = r['main_loop'].get('mfma_busy')
    else:
        out['roofline'] = None
    out['cpu_baseline'] = full.get('cpu_baseline')
    par = full.get('parity')
    if par:
        out['parity'] = {'tol': {'rtol': PARITY_TOL, 'atol': dict(PARITY_ATOL), 'atol_other_scalars': 0.0, 'counting_stats_abs': dict(COUNT_TOL_STATE),
                                 'form': '|x - ref| <= max(rtol * |ref|, atol) on every term of the loss',
                                 'kl': {'fresh_rtol': KL_TOL['fresh'], 'stress_rtol': KL_TOL['stress'],
                                        'why': 'enters no loss / gradient (adaptive-schedule thresholds at factors of 2); in the stress '
                                               'state (kl ~ 1) the 2^-12 rounding of 16-bit weight shadows is a systematic error of mu'}},
                         'reference': 'oracle/restated.py (f32, host) on identical inputs, full-size step',
                         'fresh': _parity_scalars(par.get('fresh')), 'stress': _parity_scalars(par.get('stress')),
                         'headline_ok': full.get('headline_parity_ok')}
    else:
        out['parity'] = None
    q = full.get('qualifying_mode')
    if q:
        out['qualifying_mode'] = {k: q.get(k) for k in ('precision', 'value', 'unit', 'ms_per_step', 'fresh_max_loss_rel',
                                                        'stress_max_loss_rel', 'stress_ok', 'fresh_worst_grad_rel_l2')}
        out['qualifying_mode']['is_headline'] = q.get('precision') == (full['config'].get('precision_mode') or '').split(':')[0]
    else:
        out['qualifying_mode'] = None
    t = full.get('throughput_mode')
    out['throughput_mode'] = {k: t.get(k) for k in ('precision', 'value', 'unit', 'ms_per_step')} if t else None
    st = full.get('strict_mode')
    out['strict_mode'] = {k: st.get(k) for k in ('precision', 'value', 'unit', 'ms_per_step', 'ok', 'fresh_max_true_rel', 'fresh_scalar',
                                                 'stress_max_true_rel', 'stress_scalar', 'fresh_worst_grad_rel_l2')} if st else None
    out['fallthrough'] = [{k: f.get(k) for k in ('precision', 'value', 'fresh_ok', 'stress_ok')} | {'stress_worst': f['stress']}
                          for f in (full.get('fallthrough') or [])]
    out['dist'] = full.get('dist')
    c5 = full.get('config5_16384_envs')
    out['config5_16384_envs'] = {k: c5.get(k) for k in ('precision', 'value', 'unit', 'ms_per_step')} if c5 else None
    out['runtime'] = full.get('runtime')
    out['detail'] = detail_path
    return out



ENGINE_OPTS = {}          # --engine-opts: applied to every agent the run builds


def load_cfg():
    from ase_amd import cfg as defaults
    return defaults.get('ase')


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


class TimedBackend:
    """Wraps a HipBackend: brackets every matrix-core launch with HIP events on the launch stream."""

    def __init__(self, be):
        self._be = be
        self.records = []          # (kind, flops, start_event, end_event)
        self.hbm = []              # (kernel, algorithmic bytes, start_event, end_event)
        self.bytes = {}            # algorithmic operand bytes per NT kernel class (A + B + C [+ mask operand / mask output])

    def __getattr__(self, k):
        return getattr(self._be, k)

    @property
    def x3(self):                      # (the engine toggles it around the f32 value path of the penalty: forward, not shadow)
        return self._be.x3

    @x3.setter
    def x3(self, v):
        self._be.x3 = v

    def _timed(self, kind, flops, fn, *a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn(*a, **kw)
        e.record()
        self.records.append((kind, flops, s, e, self._shape, getattr(self, '_nbytes', 0.0)))
        self._nbytes = 0.0

    def gemm_nt(self, A, B, Cm, M, N, K, **kw):
        from ase_amd import lib as L
        self._shape = (M, N, K)
        es = A.element_size()
        aux = kw.get('aux')
        nbytes = (M * K + N * K) * es + M * N * Cm.element_size() + (aux.shape[1] * aux.element_size() * M if aux is not None else 0) \
            + (M * N // 8 if kw.get('mask_out') is not None else 0)
        # 2 = the phased 256 x 256 kernel (the dominant kernel of the update), 6 = its 192 x 256 variant (another symbol in a kernel
        # trace): each timed as its own class
        kid = self._be.lib.ase_hip_gemm_nt_kernel_id(M, N, K, self._be._gemm_code(A.dtype))
        # (the penalty's value path - f32 storage, three 16-bit MFMAs per product - is its own class: its flop count below is the
        #  ALGORITHMIC one, a third of what its matrix instructions execute)
        kind = 'nt8' if kid == 2 else 'nt8_192' if kid == 6 else ('nt_x3' if (A.dtype == torch.float32 and self._be.x3) else 'nt')
        self.bytes[kind] = self.bytes.get(kind, 0.0) + nbytes
        self._nbytes = float(nbytes)
        self._timed(kind, 2.0 * M * N * K, self._be.gemm_nt, A, B, Cm, M, N, K, **kw)

    def gemm_tn(self, A, B, G, M, N, K, n_real, k_real, split_src, split_dst, **kw):
        self._shape = (M, N, K)
        self._timed('tn', 2.0 * M * n_real * k_real, self._be.gemm_tn, A, B, G, M, N, K, n_real, k_real, split_src,
                    split_dst, **kw)

    def gemm_tn_grouped(self, plan):
        flops = sum(2.0 * M * nr * kr for (A, B, G, gb, br, M, N, K, nr, kr, ss, sd, al) in plan['keep'])
        self._shape = (0, len(plan['keep']), plan['n_work'])       # grouped: N = problems, K = work items
        es = plan['keep'][0][0].element_size()        # both operands once + the f32 gradient read-modify-write
        self._nbytes = float(sum(M * (N + K) * es + 8 * nr * kr for (A, B, G, gb, br, M, N, K, nr, kr, ss, sd, al) in plan['keep']))
        self._timed('tn', flops, self._be.gemm_tn_grouped, plan)

    # ---- HBM-bound kernels: algorithmic bytes per launch, HIP events around the launch
    def _hbm(self, name, nbytes, fn, *a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn(*a, **kw)
        e.record()
        self.hbm.append((name, float(nbytes), s, e))

    def rms_moments(self, src, D, idx, remap, M, state, sums):
        self._hbm('rms_moments (obs)' if D < 1000 else 'rms_moments', M * D * 4, self._be.rms_moments, src, D, idx, remap, M, state, sums)

    def rms_moments_multi(self, streams, D, M, state, sums_list):
        self._hbm('rms_moments_multi (3 amp streams)', len(streams) * M * D * 4, self._be.rms_moments_multi, streams, D, M, state, sums_list)

    def rms_normalize(self, src, D, idx, remap, M, mean, std, outs):
        wb = sum(o.element_size() for o in outs if o is not None)
        self._hbm('rms_normalize', M * D * (4 + wb), self._be.rms_normalize, src, D, idx, remap, M, mean, std, outs)

    def rms_normalize_multi(self, streams, D, M, means, stds, outs):
        self._hbm('rms_normalize_multi (3 amp streams)', len(streams) * M * D * (4 + outs[0].element_size()),
                  self._be.rms_normalize_multi, streams, D, M, means, stds, outs)

    def gather_multi(self, desc, items, idx, remap, M):
        nb = sum(M * D * (4 + dst.element_size()) for (src, D, dst) in items)
        self._hbm('gather_multi', nb, self._be.gather_multi, desc, items, idx, remap, M)

    def apply_multi(self, desc, items, dtype, opt_state, acc):
        n = sum(it[0].numel() + it[5].numel() for it in items)
        # per parameter: read w, g, m, v (16 B), write w, m, v (12 B), write the two compute-dtype shadows (2 x 2 B)
        self._hbm('apply_multi', n * 32, self._be.apply_multi, desc, items, dtype, opt_state, acc)

    def ring_store(self, src, D, idx, remap, n, dst, size, head):
        self._hbm('ring_store', n * D * 8, self._be.ring_store, src, D, idx, remap, n, dst, size, head)

    def hbm_summary(self, peak_gbps=8000.0):
        torch.cuda.synchronize()
        agg = {}
        for name, nb, s, e in self.hbm:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += nb
            a[2] += s.elapsed_time(e)
        return [{'kernel': k, 'launches': v[0], 'algorithmic_bytes_per_launch': round(v[1] / v[0]), 'avg_us': round(v[2] * 1e3 / v[0], 2),
                 'GBps': round(v[1] / (v[2] * 1e-3) / 1e9, 1), 'frac_of_8TBps': round(v[1] / (v[2] * 1e-3) / 1e9 / peak_gbps, 4)}
                for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])]

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind in ('nt8', 'nt8_192', 'nt', 'nt_x3', 'tn'):
            rs = [r for r in self.records if r[0] == kind]
            if rs:
                ms = sum(r[2].elapsed_time(r[3]) for r in rs)
                out[kind] = {'launches': len(rs), 'ms': ms, 'flops': sum(r[1] for r in rs)}
        return out

    def breakdown(self, peak_tflops=2500.0, peak_gbps=8000.0):
        """Per-shape table: HIP-event time, TF/s, and the shape's own roofline - floor = max(flop / MFMA peak, algorithmic
        operand bytes / 8 TB/s); `of floor` = floor / measured (event brackets include ~4-8 us of launch gap)."""
        torch.cuda.synchronize()
        agg = {}
        for kind, flops, s, e, shape, nb in self.records:
            a = agg.setdefault((kind,) + shape, [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += s.elapsed_time(e)
            a[2] += flops
            a[3] += nb
        rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
        out = []
        for k, v in rows:
            us = v[1] * 1e3 / v[0]
            t_m, t_h = v[2] / v[0] / (peak_tflops * 1e12) * 1e6, v[3] / v[0] / (peak_gbps * 1e9) * 1e6
            bound = 'hbm ' if t_h > t_m else 'mfma'
            roof = f'  {v[3] / v[0] / 1e6:7.1f} MB  floor {max(t_m, t_h):6.1f} us ({bound})  {max(t_m, t_h) / us:5.2f} of floor' if v[3] else ''
            out.append(f'{k[0]} M={k[1]:6d} N={k[2]:5d} K={k[3]:5d}  n={v[0]:4d}  {v[1]:8.2f} ms  {us:7.1f} us  '
                       f'{v[2] / (v[1] * 1e-3) / 1e12:7.1f} TF/s{roof}')
        return out


def sustained_clock(be, device, dtype, M=16384, N=1024, K=1024, reps=40):
    """Clock frequency the chip sustains under the dominant kernel (not its 2.4 GHz boost): `reps` back-to-back launches of the
    dominant layer shape with the kernel's debug stamps switched on (ase_hip_debug_nt_profile) - once with the main-loop stamps
    in the 100 MHz real-time clock, once in shader clocks (s_memtime); MHz = cycles / us, averaged over the workgroups of the
    LAST launch.  Returns (MHz, main-loop us, main-loop shader cycles) or None if the shape is not on the phased kernel."""
    from ase_amd import lib as L
    if be.lib.ase_hip_gemm_nt_kernel_id(M, N, K, be._gemm_code(dtype)) != 2:
        return None
    g = torch.Generator(device='cpu').manual_seed(7)
    A = torch.randn(M, K, generator=g).to(device=device, dtype=dtype)
    W = (torch.randn(N, K, generator=g) * 0.03).to(device=device, dtype=dtype)
    Cm = torch.empty(M, N, device=device, dtype=dtype)
    bias = torch.zeros(N, device=device, dtype=torch.float32)
    nwg = (M // 256) * (N // 256)
    prof = torch.zeros(nwg * 4, device=device, dtype=torch.int64)
    out = []
    for mode in (0, 1):
        be.lib.ase_hip_debug_nt_profile_clock(mode)
        be.lib.ase_hip_debug_nt_profile(prof.data_ptr())
        for _ in range(reps):
            be.gemm_nt(A, W, Cm, M, N, K, bias=bias, act=L.ACT_RELU)
        torch.cuda.synchronize()
        be.lib.ase_hip_debug_nt_profile(None)
        be.lib.ase_hip_debug_nt_profile_clock(0)
        t = prof.view(nwg, 4).cpu()
        out.append(float((t[:, 2] - t[:, 1]).double().mean()))
    us, cyc = out[0] / 100.0, out[1]
    return round(cyc / us, 1), round(us, 2), round(cyc)


def algorithmic_flops_per_step(eng):
    """2*M*N*K over the real (unpadded) layer shapes, forward + data-gradient + weight-gradient, the
    gradient-penalty chain included; the shared-trunk encoder costs only its head (SURVEY §8d 'minimal')."""
    M, AMB, Ra = eng.M, eng.AMB, eng.Ra

    def chain(layers, rows, first_dgrad=False):
        f = 0.0
        for i, d in enumerate(layers):
            n = sum(p[1] for p in d.parts)
            f += 2.0 * rows * n * d.K * (3 if (i > 0 or first_dgrad) else 2)
        return f
    f = chain(eng.style, Ra) + chain(eng.actor + [eng.mu_head], Ra) + 2.0 * Ra * eng.z * eng.actor[0].N   # style-column dgrad
    f += chain(eng.critic + [eng.value_head], M)
    if eng.has_disc:
        f += chain(eng.disc, 3 * AMB) + 2.0 * 3 * AMB * 1 * eng.disc_head.K * 3
        if eng.has_enc and not eng.enc_sep:
            f += 2.0 * AMB * eng.z * eng.disc_head.K * 3
        gp = sum(2.0 * AMB * d.N * d.K for d in eng.disc)
        f += gp * 3 - 0        # chain forward + its backward (data + weight)
    return f


def make_agent(device, precision, use_graph, world, rank, seed=0, force_dist=False, multi_stream=True, dp_mode='shard', engine_opts=None, extra_cfg=None,
               num_envs=4096):
    from ase_amd.learning import agents, models
    from ase_amd.learning.network_builder import ASEBuilder
    from ase_amd.synthetic import EnvSpec, SyntheticSource
    import types
    net_p, cfg = load_cfg()
    spec = EnvSpec(num_envs=num_envs, horizon=cfg['horizon_length'], obs_size=253, act_size=31, amp_obs_size=1400,
                   latent_dim=cfg['latent_dim'], latent_steps_min=cfg['latent_steps_min'],
                   latent_steps_max=cfg['latent_steps_max'])
    torch.manual_seed(seed)
    b = ASEBuilder()
    b.load(net_p)
    sp = lambda n: types.SimpleNamespace(shape=(n,))
    # 'horovod' (weak scaling): every rank owns its own 4096 environments - a different synthetic stream per rank
    src = SyntheticSource(spec, seed=1234 + 2 + (0 if dp_mode == 'shard' else 1009 * rank))
    cfg = dict(cfg)
    cfg.update(network=models.ModelASEContinuous(b), num_actors=spec.num_envs, device=device, precision=precision,
               graph_capture=use_graph, world_size=world, rank=rank, vec_env=src, force_dist=force_dist,
               multi_stream=multi_stream, dp_mode=dp_mode, engine_opts=dict(engine_opts or ENGINE_OPTS),
               env_info={'observation_space': sp(253), 'action_space': sp(31), 'amp_observation_space': sp(1400)})
    cfg.update(extra_cfg or {})
    return agents.ASEAgent('bench', cfg), cfg, spec


def _rms_dict(vec):
    D = (vec.numel() - 1) // 2
    v = vec.detach().cpu()
    return {'mean': v[:D].clone(), 'var': v[D:2 * D].clone(), 'count': v[2 * D].clone()}


class _EngineMaskedRelu:
    """`act` callable for oracle/restated.py: ReLU with the ENGINE's 0/1 derivative masks, consumed in the oracle's call order
    (y = x * mask: the value differs from relu(x) only where the two forwards disagree about a sign, i.e. |x| of the order of the
    engine's rounding error; the DERIVATIVE is the engine's everywhere).  Diagnostic only - see `grad_at_engine_masks`."""

    def __init__(self, masks):
        self.masks, self.i, self.flipped, self.total = masks, 0, 0.0, 0.0

    def __call__(self, x):
        m = self.masks[self.i]
        self.i += 1
        assert m.shape == x.shape, (self.i, m.shape, x.shape)
        self.flipped += float((m != (x > 0).to(m.dtype)).sum())
        self.total += float(m.numel())
        return x * m


def engine_relu_masks(eng):
    """The ReLU derivative masks of the engine's LAST step (its bit-mask twins) as float 0/1 tensors on the host, in the order
    oracle/restated.calc_gradients('ase', ...) calls its activation: actor (style MLP, dense layers), critic, discriminator on
    agent / replay / demo rows, encoder (the agent rows' trunk again), diversity pass (style MLP, dense layers on the new latents).
    None when the net is not the all-ReLU shared-trunk ASE net with bit masks."""
    from ase_amd import lib as L
    if eng.kind != 'ase' or eng.enc_sep or not eng._use_bits or not eng.div_on:
        return None
    layers = eng.style[:-1] + eng.actor + eng.critic + eng.disc
    if any(d.act != L.ACT_RELU for d in layers):
        return None
    M, AMB = eng.M, eng.AMB

    def unpack(h, d, r0, r1):
        bits = eng._mask_of(h)
        w = bits[r0:r1].to(torch.int64) & 0xFFFFFFFF
        b = ((w.unsqueeze(-1) >> torch.arange(32, device=w.device)) & 1).reshape(w.shape[0], -1)[:, :d.N]
        return b.to(torch.float32).cpu()
    out = []
    for r0 in (0,):
        out += [unpack(h, d, r0, r0 + M) for h, d in zip(eng.Hs, eng.style[:-1])]
        out += [unpack(h, d, r0, r0 + M) for h, d in zip(eng.Ha, eng.actor)]
    out += [unpack(h, d, 0, M) for h, d in zip(eng.Hc, eng.critic)]
    for blk in (0, 1, 2, 0):                                    # agent, replay, demo, encoder (= agent rows)
        out += [unpack(h, d, blk * AMB, (blk + 1) * AMB) for h, d in zip(eng.Hd4, eng.disc)]
    out += [unpack(h, d, M, 2 * M) for h, d in zip(eng.Hs, eng.style[:-1])]
    out += [unpack(h, d, M, 2 * M) for h, d in zip(eng.Ha, eng.actor)]
    return out


def cpu_baseline_and_parity(agent, cfg, steps=8, mode='bf16', traj_steps=None, state=None, with_times=False):
    """The reference's arithmetic (oracle/restated.py, f32, torch CPU threads = host cores) on a bounded sample of the SAME
    workload: `steps` full-size optimisation steps (minibatch 16384 / amp 4096, median step time), extrapolated to the 48
    steps of one update (the once-per-epoch tail is < 2 % and left out).

    Parity: the FIRST of those steps is also executed by the GPU engine on identical inputs - same weights (whatever the
    timed updates left), same running statistics, same minibatch rows, same demo rows, same diversity latents - and every
    reported loss scalar and every gradient tensor is compared with the oracle's.  Trajectory: both sides then take their
    OWN Adam step (fresh optimizer state on both) and go on to the next minibatch - `traj_steps` (default: all `steps`)
    consecutive optimisation steps, loss scalars compared at every one of them (errors of the weights compound)."""
    from oracle import restated as R
    from ase_amd import lib as L
    ncpu = host_cores()
    torch.set_num_threads(ncpu)
    eng = agent.engine
    dev = eng.dev
    B, MB, AMB = agent.batch_size, agent.minibatch_size, cfg['amp_minibatch_size']
    H, N = agent._remap
    env_major = lambda t: t.view(H, N, -1).transpose(0, 1).reshape(H * N, -1)
    ds = {k: env_major(v).cpu() for k, v in agent._ds.items()}
    for k in ('old_logp_actions', 'advantages', 'rand_action_mask'):
        ds[k] = ds[k].view(-1)
    ds['amp_obs_replay'] = ds['amp_obs']
    g = torch.Generator().manual_seed(0)
    demo = agent._amp_obs_demo_buffer.data
    dsel = torch.randint(0, demo.shape[0], (B,), generator=g)
    ds['amp_obs_demo'] = demo.cpu()[dsel]
    sd = R.canonical_sd(agent.model.state_dict(), False,
                        requires_grad=[k.replace('a2c_network.', '', 1) for k, p in agent.model.named_parameters() if p.requires_grad])
    sd = {k: (v.cpu().detach().requires_grad_(True) if v.requires_grad else v.cpu()) for k, v in sd.items()}
    rms = {'obs': _rms_dict(eng.obs_state), 'amp': _rms_dict(eng.amp_state)}
    adam = R.adam_new()
    perm = torch.randperm(B, generator=g)
    times, parity = [], None
    n_traj = steps if traj_steps is None else min(steps, traj_steps)
    # fresh optimizer state on the GPU side too (the oracle's is new): first / second moments and the step counter
    eng.adam_m.zero_()
    eng.adam_v.zero_()
    eng.opt_state[0] = 0.0
    counts = ('actor_clip_frac', 'disc_agent_acc', 'disc_demo_acc')
    scale = {k: a / PARITY_TOL for k, a in PARITY_ATOL.items()}     # the atol as a floor on |ref|: see PARITY_ATOL
    traj = []
    for i in range(steps):
        pos = i % (B // MB)
        idx = perm[pos * MB:(pos + 1) * MB]
        mb = {k: v[idx] for k, v in ds.items()}
        z = R.sample_latents(MB, 64, g)
        if i < n_traj:
            # ---- the GPU engine on the same step (its running statistics advance like the oracle's)
            idx_d = idx.to(torch.int32).to(dev)
            arows = idx_d[:AMB].contiguous()
            streams = [(agent._ds['amp_obs'], arows, agent._remap), (agent._ds['amp_obs'], arows, agent._remap),
                       (demo, dsel[idx[:AMB]].to(torch.int32).to(dev), (0, 0))]
            eng.step(agent._ds, idx_d, agent._remap, streams, new_z=z.to(dev), apply=False)
            torch.cuda.synchronize()
            res_g = {k: v.detach().cpu().clone() for k, v in eng.results().items()}
            if i == 0:
                grads_g = {k: v.detach().cpu().clone() for k, v in eng.export_grads().items()}
                masks_g = engine_relu_masks(eng)
                rms0 = {k: {kk: vv.clone() for kk, vv in v.items()} for k, v in rms.items()}
            # its optimizer step (the weight-only loss terms are in the gradients already): step counter, Adam, shadows
            eng.be.begin_step(eng.opt_state, None)
            eng.be.adam(eng.params[:eng.n_train], eng.grads[:eng.n_train], eng.adam_m[:eng.n_train], eng.adam_v[:eng.n_train],
                        eng.opt_state)
            eng.refresh_shadows()
        t0 = time.time()
        res = R.calc_gradients('ase', sd, rms, mb, cfg, z)
        if i < n_traj:
            loss_rel, count_abs, true_rel = {}, {}, {}
            for k in ('actor_loss', 'critic_loss', 'b_loss', 'entropy', 'kl', 'actor_clip_frac', 'disc_loss', 'disc_grad_penalty',
                      'disc_logit_loss', 'disc_agent_acc', 'disc_demo_acc', 'enc_loss', 'amp_diversity_loss'):
                r = float(res[k].mean())
                loss_rel[k] = abs(float(res_g[k].mean()) - r) / max(abs(r), scale.get(k, 0.0), 1e-12)
                true_rel[k] = abs(float(res_g[k].mean()) - r) / max(abs(r), 1e-12)          # no scale floor: |delta| / |ref|
                if k in counts:
                    count_abs[k] = abs(float(res_g[k].mean()) - r)
            wl_i = max((k for k in loss_rel if k not in counts), key=loss_rel.get)
            wc_i = max(counts, key=loss_rel.get)
            traj.append((loss_rel[wl_i], wl_i, loss_rel[wc_i], wc_i))
        if i == 0:
            true0 = dict(true_rel)
            grad_rel = {}
            for k, p in sd.items():
                if p.requires_grad:
                    grad_rel[k] = float((grads_g[k].double() - p.grad.double()).norm() / p.grad.double().norm().clamp_min(1e-30))
            wk = max(grad_rel, key=grad_rel.get)
            # the same comparison with the oracle's ReLU derivative masks replaced by the ENGINE's: what remains is the error of the
            # arithmetic proper.  A hidden unit whose pre-activation changes sign under the forward's storage rounding (|z| ~ 2^-12
            # of its scale in half) is an O(1) error of one element of dZ; a flipped fraction f shows as sqrt(f) in relative L2 -
            # ~1e-4 of the units, 1-3 % of the gradient norm in half (bf16: sqrt(8) x more), for ANY precision of the backward
            # (profiles/r05_grad_error_sources.txt).  Both choices are subgradients at the kink; the loss scalars above never see it.
            masked = None
            if masks_g is not None:
                sd_m = {k: (v.detach().clone().requires_grad_(True) if v.requires_grad else v) for k, v in sd.items()}
                act_m = _EngineMaskedRelu(masks_g)
                R.calc_gradients('ase', sd_m, rms0, mb, cfg, z, act=act_m)
                gm = {k: float((grads_g[k].double() - p.grad.double()).norm() / p.grad.double().norm().clamp_min(1e-30))
                      for k, p in sd_m.items() if p.requires_grad}
                wm = max(gm, key=gm.get)
                masked = {'median_grad_rel_l2': float(f'{sorted(gm.values())[len(gm) // 2]:.3e}'), 'worst_grad_rel_l2': float(f'{gm[wm]:.3e}'),
                          'worst_grad_tensor': wm, 'flipped_mask_fraction': float(f'{act_m.flipped / max(act_m.total, 1.0):.3e}'),
                          'what': 'gradient tensors against the oracle evaluated with the ENGINE\'s ReLU derivative masks (0/1 '
                                  'multipliers in place of relu\'): the arithmetic error without the sign flips of near-zero units'}
                del sd_m, masks_g
            # counting statistics (fractions of samples on one side of a threshold) move in steps of 1/rows whenever a
            # near-threshold sample flips: reported separately from the continuous loss scalars
            wl = max((k for k in loss_rel if k not in counts), key=loss_rel.get)
            wc = max(counts, key=loss_rel.get)
            parity = {'mode': mode, 'state': state,
                      'what': f'first oracle step (minibatch {MB}, amp {AMB}) re-run by the GPU engine on identical '
                                            'inputs, no optimizer step; reference = oracle/restated.py in f32 on the host',
                      'max_loss_rel': float(f'{loss_rel[wl]:.3e}'), 'max_loss_rel_scalar': wl,
                      # the gradient penalty is a cancelling sum in the discriminator's weights (training drives it small):
                      # its error is the 16-bit rounding of those weights (profiles/r03_gp_error_sources.txt) - listed apart
                      'max_loss_rel_without_grad_penalty': float(f'{max(v for k, v in loss_rel.items() if k not in counts and k != "disc_grad_penalty"):.3e}'),
                      'max_count_stat_rel': float(f'{loss_rel[wc]:.3e}'), 'max_count_stat': wc,
                      'max_count_stat_abs': float(f'{max(count_abs.values()):.3e}'),
                      'loss_rel': {k: float(f'{v:.2e}') for k, v in loss_rel.items()},
                      # the same errors divided by |reference value| alone: actor_loss and enc_loss are means of signed O(1)
                      # summands (normalised advantages have zero mean), their VALUE is a small remainder of the sum
                      'loss_true_rel': {k: float(f'{v:.2e}') for k, v in true0.items() if k not in counts},
                      'max_loss_true_rel': float(f'{max(v for k, v in true0.items() if k not in counts):.3e}'),
                      'max_loss_true_rel_scalar': max((k for k in true0 if k not in counts), key=true0.get),
                      'worst_grad_rel_l2': float(f'{grad_rel[wk]:.3e}'), 'worst_grad_tensor': wk,
                      'median_grad_rel_l2': float(f'{sorted(grad_rel.values())[len(grad_rel) // 2]:.3e}'),
                      'grad_at_engine_masks': masked}
        R.adam_step(sd, adam, cfg['learning_rate'])
        times.append(time.time() - t0)
    if parity is not None and traj:
        worst = max(range(len(traj)), key=lambda j: traj[j][0])
        parity['trajectory'] = {
            'what': f'{len(traj)} consecutive optimisation steps, GPU engine and oracle each with its own Adam (fresh state), same '
                    'minibatches / demo rows / diversity latents; worst continuous loss scalar and worst counting statistic per step',
            'steps': len(traj), 'max_loss_rel': float(f'{traj[worst][0]:.3e}'), 'max_loss_rel_scalar': traj[worst][1],
            'max_loss_rel_step': worst, 'max_count_stat_rel': float(f'{max(t[2] for t in traj):.3e}'),
            'per_step_max_loss_rel': [float(f'{t[0]:.2e}') for t in traj]}
    if with_times:
        return times, parity
    return cpu_from_times(times, cfg, B, MB, AMB, ncpu), parity


def cpu_from_times(times, cfg, B, MB, AMB, ncpu):
    tt = sorted(times[1:]) if len(times) > 1 else times           # first call = warm-up
    t_step = tt[len(tt) // 2]
    n_steps = cfg['mini_epochs'] * (B // MB)
    cpu = {'value': B / (n_steps * t_step), 'unit': 'samples/s', 'cores': ncpu, 'kind': 'port',
           'sample': f'{len(tt)} of the {n_steps} optimisation steps of one update at full size (minibatch {MB}, amp {AMB}; one '
                     f'more as warm-up), median {t_step:.2f} s/step on {ncpu} threads, extrapolated x{n_steps}; '
                     'oracle/restated.py (f32 torch CPU)'}
    # the UNMODIFIED reference cannot travel to the GPU box (/root/reference is not there): it was timed beside this port in the
    # authoring container, same inputs, same process (oracle/time_reference.py) - the port is the faster of the two
    try:
        j = json.load(open(os.path.join(ROOT, 'profiles', 'r04_reference_cpu_timing.json')))
        cpu['reference_beside_port'] = {'measured_in_this_run': False, 'port_over_reference_time': j['port_over_reference'], 'cores': j['cores'],
                                        'reference_s_per_step': j['reference']['s_per_step'], 'port_s_per_step': j['port']['s_per_step'],
                                        'source': 'profiles/r04_reference_cpu_timing.json (oracle/time_reference.py, authoring container)'}
    except (OSError, KeyError, ValueError):
        pass
    return cpu


def fill_rollout(agent, device):
    """Untimed: a fresh synthetic rollout into HBM.  The policy outputs (mu, sigma, value -> actions, neglogp) come from the
    engine's OWN inference path in the agent's precision, with the agent's current weights: the importance ratio of the
    first optimisation step on this rollout is ~1, as in real training."""
    with torch.no_grad():
        agent.set_eval()
        exp = agent.vec_env.experience(agent._cpu_policy())
        for k, v in exp.items():
            if k in agent.experience:
                agent.experience[k].copy_(v.to(device))
    torch.cuda.synchronize()


def parity_both_states(agent, cfg, device, mode, steps_fresh, steps_stress, stale_updates=2):
    """Parity of one precision mode where training lives and in the corner the timed loop ends in:
      fresh  - a NEW rollout produced with the current weights (after at least one update), first optimisation step on it:
               importance ratio ~ 1, clip fraction ~ 0 - every step of real training starts here;
      stress - the same rollout after `stale_updates` more full updates (96 optimisation steps) on it: the policy is far
               from the behaviour policy (clip fraction ~ 0.9), the actor gradient is a small cancelling remainder.
    Returns (oracle step times, {'fresh': ..., 'stress': ...})."""
    fill_rollout(agent, device)
    agent._play_steps_tail()
    t1, fresh = cpu_baseline_and_parity(agent, cfg, steps=steps_fresh, mode=mode, state='fresh rollout, first step', with_times=True)
    for _ in range(stale_updates):
        info = agent.update(agent._play_steps_tail())
    agent._play_steps_tail()
    t2, stress = cpu_baseline_and_parity(agent, cfg, steps=steps_stress, mode=mode,
                                         state=f'stale rollout after {stale_updates} more updates on it', with_times=True)
    stress['train_result_before'] = {k: round(float(v[-1]), 4) for k, v in info.items()
                                     if k in ('kl', 'actor_clip_frac') and torch.is_tensor(v[-1])}
    return t1 + t2[1:], {'fresh': fresh, 'stress': stress}


def time_updates(agent, n, prime):
    """Median update time (HIP events) of n updates after `prime` untimed ones."""
    torch.set_num_threads(1)          # (the oracle leg before this one raised it: see --host-threads)
    for _ in range(prime):
        agent.update(agent._play_steps_tail())
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    marks[0].record()
    for i in range(n):
        agent.update(agent._play_steps_tail())
        marks[i + 1].record()
    torch.cuda.synchronize()
    return sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(n))[n // 2]


def _dbg(msg):
    if os.environ.get('ASE_BENCH_DEBUG'):
        print(f'[rank {os.environ.get("RANK", 0)}] {msg}', file=sys.stderr, flush=True)


HEADLINE_CANDIDATES = ['f16gpx3', 'f16gp32', 'f32']     # fastest first; the headline is the first whose parity holds IN THIS RUN
STRICT_CANDIDATES = ['bf16x3', 'f32']                    # strict mode: true |delta| / |ref| <= 1e-4 on every loss scalar, no floor


def _mode_ok(par):
    return bool(par) and _state_ok(par['fresh'], 'fresh') and _state_ok(par['stress'], 'stress')


def _strict_ok(par):
    """True relative error (no scale floor) of every continuous loss scalar within 1e-4 in both rollout states."""
    return bool(par) and all(par[s_]['max_loss_true_rel'] <= PARITY_TOL and par[s_]['max_count_stat_abs'] <= COUNT_TOL_STATE[s_]
                             for s_ in ('fresh', 'stress'))


def measure_mode(args, precision, device, world, rank, use_graph):
    """The contract's timed protocol for one precision mode: build the agent, synthetic rollout into HBM, W untimed warm-up
    updates, EXACTLY K timed updates between barrier + synchronize brackets (max over ranks), then the instrumented eager
    single-stream update for the roofline of the dominant kernel."""
    t_setup = time.time()
    agent, cfg, spec = make_agent(device, precision, use_graph, world, rank, force_dist=args.force_dist,
                                   multi_stream=not args.no_multi_stream, dp_mode=args.dp_mode)
    weak = world > 1 and args.dp_mode == 'horovod'
    Bl = agent.batch_size                                    # this rank's samples per update
    B = Bl * (world if weak else 1)                          # samples of one update over ALL ranks
    fill_rollout(agent, device)
    agent._init_amp_demo_buf()
    torch.cuda.synchronize()
    if rank == 0:
        print(f'[bench] {precision}: setup + synthetic rollout: {time.time() - t_setup:.1f} s', file=sys.stderr)
    tail_marks = []            # (event before the tail, event after it) per update: SURVEY §8d also wants the 48-step-only figure

    def one_update():
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        batch = agent._play_steps_tail()
        b.record()
        tail_marks.append((a, b))
        return agent.update(batch)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    torch.set_num_threads(max(1, args.host_threads))
    # recording the launch programs (one per optimisation step, for the "first epoch" and the "replay ring" variant of the
    # replay source) is setup, like a compile step: two untimed priming updates record everything, so that the W warm-up and
    # K timed steps below are pure replays whatever W is.
    if use_graph:
        for _ in range(2):
            one_update()
    for _ in range(args.warmup):
        one_update()
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    host_ms = []
    if args.gc == 'freeze':      # a launcher-level choice: no cyclic-GC pass inside the timed updates (see --gc)
        import gc
        gc.collect()
        gc.freeze()
        gc.disable()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        th = time.perf_counter()
        info = one_update()
        marks[i + 1].record()
        host_ms.append((time.perf_counter() - th) * 1e3)
    sync()
    dt = time.perf_counter() - t0
    if args.gc == 'freeze':
        import gc
        gc.enable()
    ms_tail = sum(a.elapsed_time(b) for a, b in tail_marks[-args.steps:]) / args.steps
    if args.verbose and rank == 0:
        try:
            print('[bench] cgroup cpu.stat after the timed updates: ' + ' '.join(open('/sys/fs/cgroup/cpu.stat').read().split()), file=sys.stderr)
        except OSError:
            pass
        print('[bench] per-update ms: ' + ' '.join(f'{marks[i].elapsed_time(marks[i + 1]):.1f}' for i in range(args.steps)),
              file=sys.stderr)
        print('[bench] host enqueue ms: ' + ' '.join(f'{h:.1f}' for h in host_ms), file=sys.stderr)
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = B * args.steps / dt
    last = {k: float(v[-1]) for k, v in info.items() if torch.is_tensor(v[-1]) and v[-1].numel() == 1}

    # ---- roofline of the dominant kernel class (matrix-core GEMMs): HIP events around every launch of one
    # additional, eager (un-graphed) update on the same stream the kernels are launched on.
    eng = agent.engine
    tb = TimedBackend(eng.be)          # every rank runs the instrumented update (it contains collectives); rank 0 reports
    eng.be = tb
    agent.use_graph = False
    ms_flag, eng.multi_stream = eng.multi_stream, False       # serial launches: clean per-kernel durations
    one_update()
    summ = tb.summary()
    if args.breakdown:
        print('\n'.join(tb.breakdown(peak_tflops=MFMA_PEAK_TFLOPS[precision])), file=sys.stderr)
    eng.be = tb._be
    eng.multi_stream = ms_flag
    agent.use_graph = use_graph
    n_opt = cfg['mini_epochs'] * (Bl // cfg['minibatch_size'])
    alg = algorithmic_flops_per_step(eng) * n_opt + 2.0 * Bl * sum(d.N * d.K for d in eng.disc) \
        + 2.0 * Bl * (1 + eng.z) * eng.disc_head.K
    gemm_ms = sum(v['ms'] for v in summ.values())
    launches = sum(v['launches'] for v in summ.values())
    peak = MFMA_PEAK_TFLOPS[precision]
    # the dominant kernel = the GEMM kernel class with the most time (16-bit modes: the phased 256 x 256 NT kernel)
    dom = max(summ, key=lambda k: summ[k]['ms'])
    dname = {'nt8': f'gemm_nt8_kernel<{DTYPE_OF[precision]}> (phased 256x256 NT: forward + data-gradient of the wide layers)',
             'nt8_192': f'gemm_nt8_kernel<{DTYPE_OF[precision]}, 192> (phased 192x256 NT)', 'nt': 'gemm_nt_kernel (NT tiles 64/128/256)', 'tn': 'gemm_tn kernels (weight gradients)',
             'nt_x3': 'gemm_nt_kernel<f32h_t> (the penalty value path: three f16 MFMAs per product)'}[dom]
    dv = summ[dom]
    achieved = dv['flops'] / (dv['ms'] * 1e-3) / 1e12
    # HBM traffic of that kernel per launch: PMC measurement committed under profiles/ (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE in separate passes, FETCH doubled per the gfx950 note); bench cannot profile itself
    # ... and only a measurement of THIS build is quoted: the file carries the sha256 of the library its passes ran
    # (scripts/make_pmc_json.py --lib), compared with the library loaded here; a profile of another build gives traffic = null
    traffic, traffic_src = None, None
    pmc = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if f.endswith('_pmc.json')) \
        if os.path.isdir(os.path.join(ROOT, 'profiles')) else []
    if pmc and precision in ('bf16', 'f16', 'f16gp32', 'f16gpx3'):
        from scripts.make_pmc_json import lib_sha256
        from ase_amd import lib as _L
        sha = lib_sha256(_L.LIB_PATH)
        for name in reversed(pmc):
            j = json.load(open(os.path.join(ROOT, 'profiles', name)))
            if j.get('kernel_class') == dom and j.get('lib_sha256') == sha:
                traffic, traffic_src = j['hbm_bytes_per_launch'], 'profiles/' + name
                break
        if traffic is None:
            traffic_src = 'none: no profiles/*_pmc.json was measured on the loaded build (sha256 %s...)' % sha[:12]
    clk = sustained_clock(eng.be, device, eng.dtype) if (dom == 'nt8' and rank == 0) else None
    roof = {'bound': 'mfma', 'kernel': dname,
            'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
            # the peak is quoted at the 2.4 GHz boost clock; what the chip holds under THIS kernel is measured in the kernel
            # (shader-clock vs real-time stamps around its main loop): frac_at_sustained_clock prices the same achieved rate
            # against peak x sustained / 2400, main_loop_mfma_busy = the matrix pipe's share of the main loop's CYCLES
            'sustained_clock_mhz': clk[0] if clk else None,
            'frac_at_sustained_clock': round(achieved / (peak * clk[0] / 2400.0), 4) if clk else None,
            'main_loop': {'shape': '16384 x 1024 x 1024 (ReLU forward)', 'us': clk[1], 'shader_cycles': clk[2],
                          'mfma_cycles': 16 * 2048, 'mfma_busy': round(16 * 2048 / clk[2], 4)} if clk else None,
            'traffic': traffic, 'traffic_unit': 'HBM bytes per launch of that kernel (PMC)', 'traffic_source': traffic_src,
            'launches': dv['launches'], 'avg_launch_us': round(dv['ms'] * 1e3 / dv['launches'], 2),
            'algorithmic_flop_per_launch': round(dv['flops'] / dv['launches']),
            'algorithmic_bytes_per_launch': round(tb.bytes.get(dom, 0.0) / dv['launches']) if dom in tb.bytes else None,
            'share_of_gemm_time': round(dv['ms'] / gemm_ms, 3),
            'all_gemm': {'achieved': round(alg / (gemm_ms * 1e-3) / 1e12, 2), 'frac': round(alg / (gemm_ms * 1e-3) / 1e12 / peak, 4),
                         'launches_per_update': launches, 'gemm_ms_per_update': round(gemm_ms, 3),
                         'algorithmic_tflop_per_update': round(alg / 1e12, 3)},
            'per_kind': {k: {'launches': v['launches'], 'ms': round(v['ms'], 3),
                             'tflops': round(v['flops'] / (v['ms'] * 1e-3) / 1e12, 1)} for k, v in summ.items()},
            'hbm_kernels': tb.hbm_summary()}
    return {'precision': precision, 'agent': agent, 'cfg': cfg, 'value': value, 'ms_per_step': ms_per_step, 'ms_tail': ms_tail,
            'last': last, 'roofline': roof, 'B': B, 'Bl': Bl, 'weak': weak,
            'grad_scale': agent.engine.gs}


def main():
    import faulthandler
    faulthandler.enable()
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--precision', default='auto',
                    help="headline precision mode.  'auto' (default): the FASTEST of f16gpx3 > f16gp32 > f32 whose loss scalars hold the "
                         '1e-4 parity bar on a fresh rollout AND in the stale-rollout stress state IN THIS RUN - a candidate that misses '
                         'is reported under `fallthrough` and the next one is measured with the same timed protocol (N > 1 or '
                         '--no-cpu-baseline: no oracle leg, the first candidate is taken).  A named mode is measured as given.',
                    choices=['auto', 'bf16', 'f16', 'f16gp32', 'f16gpx3', 'f32', 'bf16x3'])
    ap.add_argument('--no-graph', action='store_true', help='eager launches from Python (no recorded launch program)')
    ap.add_argument('--hipgraph', action='store_true', help='replay captured hipGraphs instead of the library launch programs')
    ap.add_argument('--no-multi-stream', action='store_true', help='launch the three network branches on ONE stream')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=8)
    ap.add_argument('--modes', default='', help='MORE precision modes whose parity (fresh rollout + stale-rollout stress state) and '
                    'throughput go into the detail file beside the headline (comma list of bf16,f16,f16gpx3,f16gp32,f32,bf16x3; '
                    'each costs ~20-60 s)')
    ap.add_argument('--throughput-mode', default='bf16', help='a second mode timed (no parity leg) and reported as throughput_mode; "" or none = skip')
    ap.add_argument('--no-strict-mode', dest='strict', action='store_false',
                    help='skip the strict-mode block (the fastest of bf16x3 / f32 whose TRUE relative error - no scale floor - is '
                         'within 1e-4 on every loss scalar in both states; ~15-25 s)')
    ap.add_argument('--detail', default=os.path.join(ROOT, 'gpurun_out', 'bench_detail.json'),
                    help='side file for everything that does not fit the compact stdout line ("" = stderr only)')
    ap.add_argument('--no-parity-mode', action='store_true', help='same as --modes ""')
    ap.add_argument('--force-dist', action='store_true', help='run the collectives even with one rank (RCCL smoke)')
    ap.add_argument('--dp-mode', default='shard', choices=['horovod', 'shard'],
                    help='N > 1: horovod = the reference\'s own multi-GPU semantics (rl_games HorovodWrapper, learning/common_agent.py:'
                         '94-107): every rank owns 4096 environments and its own 16384-row minibatches, gradients averaged by one RCCL '
                         'all-reduce per branch and step - WEAK scaling, per-GPU work fixed; shard = the same 4096 environments with '
                         'every minibatch row-sharded over the ranks (the R-rank update equals the 1-rank update) - STRONG scaling')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'])
    ap.add_argument('--breakdown', action='store_true', help='per-shape GEMM time table on stderr')
    ap.add_argument('--verbose', action='store_true', help='per-update times on stderr')
    ap.add_argument('--host-threads', type=int, default=1,
                    help='torch CPU threads during the GPU-timed part (the CPU oracle leg sets its own): every intra-op parallel '
                         'region leaves its OpenMP workers spinning, and on a box whose cgroup quota is smaller than the core count '
                         'torch sees (16 of 256 here) that gets the whole process throttled for the rest of the CFS period')
    ap.add_argument('--no-config5', dest='config5', action='store_false', help='skip the 16384-environment batch (BASELINE configs[4] '
                    'on one GPU) timed after everything else at N = 1')
    ap.add_argument('--main-priority', type=int, default=0, help='run everything on a non-default stream of this HIP priority '
                    '(-1 = high): the engine\'s main stream is whatever stream is current')
    ap.add_argument('--engine-opts', default='', help='JSON dict of UpdateEngine.engine_opts overrides (schedule A/Bs), e.g. '
                    '\'{"xstep": false}\'')
    ap.add_argument('--gc', default='freeze', choices=['on', 'freeze'],
                    help="'freeze' (default; the agents' config['manual_gc']): gc.freeze() + gc.disable() around the timed updates (no "
                         "generation-2 pass of Python's collector inside an update); 'on': leave the collector alone")
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = f'cuda:{local}'
    if args.main_priority != 0:
        torch.cuda.set_stream(torch.cuda.Stream(device=device, priority=args.main_priority))
    dist_info = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device(device))      # RCCL over xGMI
        else:
            dist.init_process_group(args.dist_backend)                            # (test rigs without one GPU per rank)
        # what the collective library itself saw (the driver checks that RCCL carried N ranks): every rank contributes 1 to a
        # SUM all-reduce on the device; the distinct GPUs behind the ranks come from an all-gather of their PCI bus ids
        one = torch.ones(1, device=device if args.dist_backend == 'nccl' else 'cpu')     # (gloo rigs: staged through the host)
        dist.all_reduce(one)
        ids = [None] * dist.get_world_size()
        dist.all_gather_object(ids, torch.cuda.get_device_properties(local).name + ':' + str(getattr(torch.cuda.get_device_properties(local), 'pci_bus_id', local)))
        dist_info = {'dist_backend': dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else ''),
                     'world_size': dist.get_world_size(), 'ranks_counted_by_allreduce': int(one.item()),
                     'distinct_devices': len(set(ids))}

    use_graph = False if args.no_graph else ('hipgraph' if args.hipgraph else 'program')
    if args.throughput_mode in ('none', '""', "''"):
        args.throughput_mode = ''
    if args.engine_opts:
        ENGINE_OPTS.update(json.loads(args.engine_opts))
    oracle_leg = world == 1 and not args.no_cpu_baseline          # (rank 0 is the only rank then)
    candidates = HEADLINE_CANDIDATES if args.precision == 'auto' else [args.precision]

    # ---- the headline: candidates in order of speed; the first whose parity holds in BOTH rollout states is the line's value
    cpu, modes, fallthrough, head = None, {}, [], None
    for ci, prec in enumerate(candidates):
        r = measure_mode(args, prec, device, world, rank, use_graph)
        agent, cfg = r['agent'], r['cfg']
        par = None
        if oracle_leg:
            n_f = (args.cpu_steps + 2) // 2 if cpu is None else 3
            n_s = (args.cpu_steps + 1 - n_f + 1) if cpu is None else 3
            times, par = parity_both_states(agent, cfg, device, prec, steps_fresh=n_f, steps_stress=n_s)
            if cpu is None:          # the oracle's step times of the first candidate's leg ARE the cpu baseline (8 full-size steps)
                cpu = cpu_from_times(times, cfg, agent.batch_size, agent.minibatch_size, cfg['amp_minibatch_size'], host_cores())
        r['parity'] = par
        modes[prec] = {'value': round(r['value'], 1), 'unit': 'samples/s', 'ms_per_step': round(r['ms_per_step'], 3),
                       'timed': f'{args.steps} updates after {args.warmup} warm-up ones (the contract\'s protocol)',
                       'grad_scale': r['grad_scale'], 'parity': par}
        ok = _mode_ok(par) if oracle_leg else True
        if ok or ci == len(candidates) - 1:
            head = r
            head['parity_ok'] = ok if oracle_leg else None
            break
        fallthrough.append({'precision': prec, 'value': round(r['value'], 1), 'ms_per_step': round(r['ms_per_step'], 3),
                            'fresh_ok': _state_ok(par['fresh'], 'fresh'), 'stress_ok': _state_ok(par['stress'], 'stress'),
                            'fresh': {'max_loss_rel': par['fresh']['max_loss_rel'], 'scalar': par['fresh']['max_loss_rel_scalar']},
                            'stress': {'max_loss_rel': par['stress']['max_loss_rel'], 'scalar': par['stress']['max_loss_rel_scalar']},
                            'why': 'missed the 1e-4 bar in this run; the next candidate was measured with the same protocol'})
        if rank == 0:
            print(f'[bench] {prec} missed the parity bar in this run (fresh {par["fresh"]["max_loss_rel"]:.2e} '
                  f'{par["fresh"]["max_loss_rel_scalar"]}, stress {par["stress"]["max_loss_rel"]:.2e} '
                  f'{par["stress"]["max_loss_rel_scalar"]}): falling through', file=sys.stderr)
        del agent, r
        torch.cuda.empty_cache()
    precision = head['precision']
    agent, cfg = head['agent'], head['cfg']
    B, Bl, weak = head['B'], head['Bl'], head['weak']

    # ---- more modes on request (parity in both states + throughput; detail file)
    if rank == 0 and oracle_leg:
        for m in ([] if args.no_parity_mode else [m for m in args.modes.split(',') if m and m not in modes]):
            ag, cfg_m, _ = make_agent(device, m, use_graph, world, rank)
            fill_rollout(ag, device)
            ag._init_amp_demo_buf()
            ms_m = time_updates(ag, 3 if m in ('f32', 'bf16x3') else 7, prime=3)
            _, par = parity_both_states(ag, cfg_m, device, m, steps_fresh=3, steps_stress=3)
            modes[m] = {'value': round(B / (ms_m * 1e-3), 1), 'unit': 'samples/s', 'ms_per_step': round(ms_m, 3),
                        'timed': 'median of 7 (f32 / bf16x3: 3) updates after 3 priming ones, same workload',
                        'grad_scale': ag.engine.gs, 'parity': par}
            del ag
            torch.cuda.empty_cache()
    qualifying = qualifying_mode(modes) if (rank == 0 and oracle_leg) else None

    # ---- the throughput mode (bf16: the storage type north_star names; it does NOT hold 1e-4) timed beside the headline
    thr = None
    if rank == 0 and world == 1 and args.throughput_mode and args.throughput_mode != precision:
        m = args.throughput_mode
        if m in modes:
            thr = {'precision': m, 'value': modes[m]['value'], 'ms_per_step': modes[m]['ms_per_step']}
        else:
            ag, _, _ = make_agent(device, m, use_graph, world, rank)
            fill_rollout(ag, device)
            ag._init_amp_demo_buf()
            ms_m = time_updates(ag, 7, prime=3)
            thr = {'precision': m, 'value': round(B / (ms_m * 1e-3), 1), 'ms_per_step': round(ms_m, 3)}
            del ag
            torch.cuda.empty_cache()
        thr['unit'] = 'samples/s'
        thr['timed'] = 'median of 7 updates after 3 priming ones, same workload'
        thr['note'] = 'outside the 1e-4 parity bar (8 significant bits: loss scalars 1e-4 ... 4e-3 off the f32 reference); reported, not the headline'

    # ---- BASELINE configs[4]'s batch (16384 envs x horizon 32 = 524288 samples, 192 optimisation steps) on ONE GPU, with the
    # reference's network: the reference has no 'every'-layer style net (its AMPStyleCatNet1 concatenates the style code in front
    # of the first dense layer only, learning/ase_network_builder.py:262-311), and 8 GPUs are the driver's to measure
    cfg5 = None
    n_opt_head = cfg['mini_epochs'] * (Bl // cfg['minibatch_size'])
    if rank == 0 and world == 1 and args.config5:
        del agent
        head['agent'] = None
        torch.cuda.empty_cache()
        ag, cfg_5, _ = make_agent(device, precision, use_graph, world, rank, num_envs=16384)
        fill_rollout(ag, device)
        ag._init_amp_demo_buf()
        ms5 = time_updates(ag, 3, prime=2)
        cfg5 = {'workload': 'BASELINE configs[4] batch: 16384 envs x horizon 32 = 524288 samples, 192 optimisation steps per update, '
                            'the reference\'s ASE network, 1 GPU', 'precision': precision, 'value': round(ag.batch_size / (ms5 * 1e-3), 1),
                'unit': 'samples/s', 'ms_per_step': round(ms5, 3), 'timed': 'median of 3 updates after 2 priming ones'}
        del ag
        torch.cuda.empty_cache()

    # (after the throughput mode and the 16384-environment batch: its f32 legs are ~30 s of the heaviest launches and leave the
    #  chip in a lower power state for whatever is timed next - measured: bf16 81.7 ms behind them, 62.8 ms in front)
    # ---- strict mode: ONE number that meets north_star's tolerance as a TRUE relative error, no scale floor, no asterisk
    strict = None
    if rank == 0 and oracle_leg and args.strict:
        tried = []
        for m in STRICT_CANDIDATES:
            if m not in modes:
                ag, cfg_m, _ = make_agent(device, m, use_graph, world, rank)
                fill_rollout(ag, device)
                ag._init_amp_demo_buf()
                ms_m = time_updates(ag, 3, prime=2)
                _, par = parity_both_states(ag, cfg_m, device, m, steps_fresh=2, steps_stress=2)
                modes[m] = {'value': round(B / (ms_m * 1e-3), 1), 'unit': 'samples/s', 'ms_per_step': round(ms_m, 3),
                            'timed': 'median of 3 updates after 2 priming ones, same workload', 'grad_scale': ag.engine.gs, 'parity': par}
                del ag
                torch.cuda.empty_cache()
            par = modes[m]['parity']
            ok = _strict_ok(par)
            tried.append({'precision': m, 'ok': ok, 'value': modes[m]['value']})
            if ok or m == STRICT_CANDIDATES[-1]:
                strict = {'precision': m, 'note': MODE_NOTE[m], 'value': modes[m]['value'], 'unit': 'samples/s',
                          'ms_per_step': modes[m]['ms_per_step'], 'ok': ok,
                          'criterion': 'TRUE |delta| / |ref| <= 1e-4 on every continuous loss scalar (no scale floor), counting '
                                       'statistics within 1e-3 absolute, fresh rollout AND stress state, in this run',
                          'fresh_max_true_rel': par['fresh']['max_loss_true_rel'], 'fresh_scalar': par['fresh']['max_loss_true_rel_scalar'],
                          'stress_max_true_rel': par['stress']['max_loss_true_rel'], 'stress_scalar': par['stress']['max_loss_true_rel_scalar'],
                          'fresh_worst_grad_rel_l2': par['fresh']['worst_grad_rel_l2'], 'tried': tried}
                break

    if world > 1 or args.force_dist:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        par = head.get('parity')
        full = {'metric': 'ASE PPO-update samples/sec (4096 envs x horizon 32)', 'value': round(head['value'], 1),
                'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(head['ms_per_step'], 3), 'higher_is_better': True,
                'scaling': 'n/a' if world == 1 else ('weak' if weak else 'strong'),
                'vs_baseline': None, 'dtype': DTYPE_OF[precision], 'data': 'synthetic',
                'config': {'workload': 'BASELINE configs[1]: ASE agent, 4096 envs x horizon 32, obs 253 / act 31 / amp obs '
                                       '1400 / latent 64, [1024,1024,512] MLPs + disc + shared-trunk encoder, minibatch 16384 '
                                       '(amp 4096) x 6 mini-epochs = 48 optimisation steps per update; random-init weights',
                           'precision_mode': precision + ': ' + MODE_NOTE[precision],
                           'precision_choice': ('auto: first of ' + ' > '.join(HEADLINE_CANDIDATES) + ' that held the parity bar in this run'
                                                if args.precision == 'auto' and oracle_leg else
                                                ('auto without an oracle leg: the first candidate' if args.precision == 'auto' else 'named on the command line')),
                           'samples_per_step': B, 'optimisation_steps_per_step': n_opt_head,
                           'ms_epoch_tail': round(head['ms_tail'], 3), 'ms_optimisation_steps_only': round(head['ms_per_step'] - head['ms_tail'], 3),
                           'replay': use_graph if use_graph else 'eager',
                           'parallelism': 'single GPU' if world == 1 else
                           (f'dp{world} horovod semantics (learning/common_agent.py:94-107): 4096 envs + a 16384-row minibatch per GPU, '
                            'gradients averaged by RCCL all-reduce, one bucket per branch' if weak else
                            f'dp{world} shard: the SAME 4096 envs, every 16384-row minibatch row-sharded over the ranks, RCCL '
                            'gradient all-reduce (sum), the R-rank update equals the 1-rank update')},
                'engine_opts': dict(ENGINE_OPTS),
                'runtime': ase_amd.hw_queue_note + ('; Python cyclic GC frozen during the timed updates' if args.gc == 'freeze' else '')
                + f'; torch CPU threads {max(1, args.host_threads)} during the GPU-timed part',
                'roofline': head['roofline'], 'cpu_baseline': cpu, 'qualifying_mode': qualifying, 'throughput_mode': thr,
                'strict_mode': strict, 'fallthrough': fallthrough, 'headline_parity_ok': head.get('parity_ok'),
                'dist': dist_info,
                'config5_16384_envs': cfg5, 'modes': modes, 'parity': par,
                'last_train_result': {k: round(v, 6) for k, v in head['last'].items()}}
        detail = write_detail(full, args.detail)
        line = compact_line(full, detail)
        sys.stdout.flush()
        try:        # RCCL's version banner sits in the C stdio buffer until exit: push it out BEFORE the JSON line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(line), flush=True)      # the ONE JSON line, last thing on stdout (< 5 KB: the driver keeps an 8 KB tail)


if __name__ == '__main__':
    main()
