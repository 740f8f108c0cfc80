"""MI355X update engine: one ASE / AMP / PPO optimisation step as an explicit sequence of HIP
kernel launches (no autograd, no allocation, no host sync) — capturable in a hipGraph.

It replaces, for the hot path, what the reference expresses as
``self.model(batch_dict)`` + the loss arithmetic + ``loss.backward()`` + ``optimizer.step()``
in ``ASEAgent.calc_gradients`` (learning/ase_agent.py:159-308 under /root/reference/ase),
``AMPAgent.calc_gradients`` (learning/amp_agent.py:266-390) and
``CommonAgent.calc_gradients`` (learning/common_agent.py:353-435), plus the rollout tail
(rewards / GAE / advantage + value normalisation).

Data layout in HBM
  * master parameters, gradients, Adam moments: four flat f32 buffers in checkpoint layout
    (one fused optimizer launch, one RCCL all-reduce over the flat gradient buffer);
  * per layer "shadow" copies in the compute dtype (bf16 or f32): W_s [P(N), Kpad] and its
    transpose Wt_s [Kpad, P(N)], zero padded (P(x) = x rounded up to 64), rewritten by the
    optimizer launch; the concat inputs [obs | z] of the first actor/critic layer are laid out as
    [obs, pad to P(obs) | z], so the latent block starts on a 128-byte boundary;
  * activations [rows, P(features)] in the compute dtype, ReLU activations with a bit-mask twin
    [rows, P(features) / 32] that the data-gradient epilogues read; head outputs and all loss math in f32;
  * the experience buffer stays time-major [H, N, ...]; minibatch rows are addressed through
    the epoch permutation (env-major flat index -> physical row) inside the gather/normalise
    kernels, so neither swap_and_flatten01 nor the dataset gather materialise anything.
"""
import math

import torch

from . import lib as L


def P(x, m=64):
    return (int(x) + m - 1) // m * m


# rl_games activations_factory names (learning/ase_network_builder.py:162) -> epilogue codes; 'swish' is SiLU
_ACT = {'relu': L.ACT_RELU, 'tanh': L.ACT_TANH, 'None': L.ACT_NONE, 'none': L.ACT_NONE, None: L.ACT_NONE,
        'swish': L.ACT_SILU, 'silu': L.ACT_SILU, 'elu': L.ACT_ELU, 'gelu': L.ACT_GELU, 'sigmoid': L.ACT_SIGMOID,
        'selu': L.ACT_SELU, 'softplus': L.ACT_SOFTPLUS}
# what the data-gradient epilogue multiplies by: ReLU - the activation's sign (bit-mask twin), tanh - 1 - y^2 of the output
# itself, the smooth activations - act'(z) of the pre-activation the forward launch kept in the layer's twin buffer
_AUX = {L.ACT_RELU: L.AUX_RELU_MASK, L.ACT_TANH: L.AUX_TANH_GRAD, L.ACT_NONE: L.AUX_NONE}
_AUX.update({a: L.AUX_PREACT | (a << 8) for a in range(L.ACT_SILU, L.ACT_SOFTPLUS + 1)})


class Dense:
    """One linear layer (or a group of heads sharing an input, stacked along N at 64-aligned offsets)."""

    def __init__(self, parts, k_in, act, split=None):
        # parts: list of (param prefix, n_real, row offset in the padded N dimension)
        self.parts = parts
        self.K = k_in
        self.act = _ACT[act] if not isinstance(act, int) else act
        if split is None:
            self.split_src = self.split_dst = k_in
            self.k_pad = P(k_in)
        else:
            self.split_src, self.split_dst, self.k_pad = split
        self.n_pad = max(off + P(n) for _, n, off in parts)
        self.N = parts[0][1]

    @property
    def name(self):
        return self.parts[0][0]


class UpdateEngine:
    """kind: 'ase' | 'amp' | 'ppo'.  sizes: rows handled by THIS rank (M, AMB) and global counts."""

    def __init__(self, kind, net, cfg, backend, *, minibatch, amp_minibatch=0, dtype=torch.bfloat16,
                 world_size=1, rank=0, infer_rows=0, dp_mode='shard', grad_scale=None):
        """dp_mode (world_size > 1):
          'shard'    every minibatch is row-sharded over the ranks; global denominators, normaliser moments summed over
                     the ranks, gradients SUM-reduced: the R-rank update equals the 1-rank update (BASELINE config 3);
          'horovod'  the reference's multi-GPU semantics (rl_games HorovodWrapper, learning/common_agent.py:94-107): every
                     rank owns its environments and minibatches, local statistics, gradients AVERAGED over the ranks,
                     running statistics averaged once per epoch (CommonAgent.sync_stats)."""
        self.kind, self.net, self.cfg, self.be = kind, net, cfg, backend
        self.dtype = dtype
        self.dev = net.flat_params.device
        self.R, self.rank = world_size, rank
        assert dp_mode in ('shard', 'horovod')
        self.shard = dp_mode == 'shard'
        div = world_size if self.shard else 1
        assert minibatch % div == 0 and amp_minibatch % div == 0
        self.Mg, self.AMBg = minibatch, amp_minibatch
        self.M, self.AMB = minibatch // div, amp_minibatch // div
        # Static gradient scale of half storage (dtype float16: what the reference's mixed_precision flag computes in -
        # torch.cuda.amp autocast + GradScaler, learning/ase_agent.py:216,271-288).  The loss heads store S * dL/d(head output);
        # every data-gradient launch is linear, so the whole back-propagated chain carries S and the weight-gradient launches
        # undo it through alpha = 1/S; bias gradients of the heads are formed unscaled inside the head kernels.  Head
        # gradients are O(1/minibatch): S = the power of two nearest minibatch/4 puts them near 1 and leaves ~2^13 of
        # headroom up to half's largest finite value and ~2^10 down to its subnormals for the deeper layers' gradients.
        # Conversions saturate (no inf), S is a power of two (exact), so unlike GradScaler nothing is skipped or adapted.
        scale_given = grad_scale is not None
        if grad_scale is None:
            grad_scale = 2.0 ** max(0, round(math.log2(max(minibatch, 4) / 4.0))) if dtype == torch.float16 else 1.0
        # loss_scale: 'static' (above) | 'dynamic' = torch.cuda.amp.GradScaler's behaviour (learning/ase_agent.py:271-288): overflow
        # detection over everything the scaled backward wrote, a SKIPPED optimizer step when it fires, backoff / growth of the scale
        # after EVERY optimisation step - all on the device (csrc/scaler.hip): the scale lives in self.scaler / self.scale_tab and the
        # launches that carry it read it there (their `*_dev` arguments), so recorded launch programs survive every change of it and
        # nothing is read back.  Default: dynamic when the configuration sets the reference's own flag (mixed_precision: True), static
        # for an explicitly named precision mode.  cfg['loss_scaler']: GradScaler's constructor arguments {init_scale 65536,
        # growth_factor 2, backoff_factor 0.5, growth_interval 2000} (powers of two).
        ls = cfg.get('loss_scale', None)
        if ls is None:
            ls = 'dynamic' if (cfg.get('mixed_precision', False) and dtype == torch.float16 and not scale_given) else 'static'
        assert ls in ('static', 'dynamic'), ls
        self.dyn_scale = ls == 'dynamic'
        if self.dyn_scale:
            assert dtype == torch.float16, "loss_scale: dynamic belongs to half storage (precision f16 / f16gp32 / f16gpx3)"
            assert not (cfg.get('graph_capture') == 'hipgraph' and (world_size > 1 or cfg.get('force_dist', False))), \
                "loss_scale: dynamic exchanges its overflow flag inside the optimizer phase: not capturable as a hipGraph"
            sc = dict(init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000)
            unknown = set(cfg.get('loss_scaler', {}) or {}) - set(sc)
            assert not unknown, f"unknown loss_scaler keys {sorted(unknown)}"
            sc.update(cfg.get('loss_scaler', {}) or {})
            for k in ('init_scale', 'growth_factor', 'backoff_factor'):
                assert sc[k] > 0 and math.log2(sc[k]) == round(math.log2(sc[k])), f"loss_scaler.{k} must be a power of two"
            assert sc['growth_factor'] >= 1.0 and sc['backoff_factor'] <= 1.0 and int(sc['growth_interval']) >= 1
            self.loss_scaler = sc
            if not scale_given:
                grad_scale = float(sc['init_scale'])
        # gs: the HOST-side (static) factor of the gradient scale, a launch argument.  Dynamic scale: 1 - the scale itself is the
        # device factor the same launches multiply in (self.scale_tab, written by _bind_params / set_grad_scale / every scaler_step)
        self._gs_init = float(grad_scale)
        self.gs = 1.0 if self.dyn_scale else float(grad_scale)
        # flags resolved once (rl_games defaults: normalize_value False, bounds_loss_coef None = no bound loss)
        # truncate_grads: global-norm clip of the whole gradient before Adam (learning/ase_agent.py:273-288): the norm needs every
        # gradient (weight-only loss terms included), so the per-branch optimizer steps give way to the end-of-step form
        self.truncate = bool(cfg.get('truncate_grads', False))
        self.grad_norm = float(cfg.get('grad_norm', 1.0))
        self.norm_value = bool(cfg.get('normalize_value', False))
        self.bounds_coef = float(cfg.get('bounds_loss_coef') or 0.0)
        # lr_schedule: adaptive (rl_games AdaptiveScheduler under the default 'legacy' schedule_type): the learning rate follows
        # every step's kl - on the device, inside the launch that forms the reported scalars
        self.adaptive_lr = cfg.get('lr_schedule', 'constant') == 'adaptive'
        self.kl_threshold = float(cfg.get('kl_threshold', 0.008))
        self.obs, self.act = net.obs_size, net.actions_num
        self.z = net.latent_dim if kind == 'ase' else 0
        self.amp = net.amp_obs_size if kind in ('amp', 'ase') else 0
        self.masked = kind in ('amp', 'ase')
        self.has_disc = kind in ('amp', 'ase')
        self.has_enc = kind == 'ase'
        self.div_on = kind == 'ase' and cfg.get('amp_diversity_bonus', 0) != 0
        self.enc_sep = bool(getattr(net, 'enc_separate', False))
        self.enc_gp = kind == 'ase' and cfg.get('enc_grad_penalty', 0) != 0 and cfg.get('enc_coef', 0) != 0
        self.mu_tanh = kind == 'ppo' and getattr(net, 'mu_tanh', False)
        # gp_f32: the gradient penalty's value path (demo-row forward, chain) in exact f32 inside a 16-bit engine (_gp_f32)
        self.gp32 = bool(cfg.get('gp_f32', False)) and self.has_disc and dtype in (torch.float16, torch.bfloat16) and \
            cfg.get('disc_coef', 0) * cfg.get('disc_grad_penalty', 0) != 0
        self._scratch = {}
        self.multi_stream = bool(cfg.get('multi_stream', True)) and getattr(backend, 'name', '') == 'hip'
        self._side_streams = None
        # Scheduling options (cfg['engine_opts'], a dict; every default is the measured best of its A/B on MI355X):
        #   tn_grouped      weight gradients of a branch queued and launched as ONE grouped grid (else one launch per layer)
        #   tn_wg_side      workgroups the planner sizes a SIDE branch's grouped launch for (it runs beside other branches)
        #   tn_early        policy weight gradients as soon as the actor's data-gradient chain is through, beside the style-MLP
        #                   tail (else ONE policy launch as the last kernel of the step; 70.4 ms either way)
        #   disc_early      head of the discriminator branch submitted at the top of the step, into the ~80 us of prologue kernels
        #   short_prologue  only the observation chain in front of the actor chain; latent copies / gathers on side streams
        #   style_early     style MLP beside the observation chain (69.3 vs 68.9 ms: the window is not idle)
        #   relu_bits       bit-mask twins of ReLU activations for the data-gradient epilogues
        #   fused_apply     weight-only loss terms + Adam + shadow refresh in one launch per branch (apply_wide: its 16-byte path)
        #   side_streams    2 = critic and discriminator on their own streams, 1 = they share one
        #   gp_scale_split  f16 mode: the gradient scale split between the two factors of the penalty chain's products (_gp_scales)
        #   xstep           the head of the discriminator branch (zero its gradient bucket, AMP moments -> normalise -> forward) is
        #                   NOT chained behind the main stream: it follows the branch's own optimizer step of the PREVIOUS
        #                   optimisation step on its stream and runs under that step's policy tail (grouped weight gradients ->
        #                   reduce -> optimizer, ~290 us with one queue busy) and this step's prologue of tiny kernels
        #   style_side      the style MLP's backward + its three weight gradients (style_wg > 0: own small grouped launch sized for
        #                   that many workgroups; 0: direct launches of the split-M kernel) beside the policy's wide
        #                   weight-gradient launch instead of in front of it.  Measured SLOWER (f16gpx3 73.9 -> 76.7 / 79.0 ms with
        #                   64 / 32 workgroups: three narrow problems over 32768 rows on a few CUs outlast the wide launch): off
        #   prefetch        (with xstep) the step's weight-independent prologue - observation moments -> running statistics ->
        #                   normalised [obs | latent] inputs, latent copies, diversity draw, loss-head fields - is submitted at the top
        #                   of the step on the critic's stream WITHOUT waiting for the main stream: it runs under the previous step's
        #                   policy tail, the inputs it writes are double-buffered by step parity (Xa / Xc / Zs)
        #   disc_after_style  (with xstep) the discriminator head's matrix launches wait until the main stream has launched the style
        #                   MLP's forward (its statistics / normalisation half stays un-chained).  Measured SLOWER (bf16 62.9 ->
        #                   65.2 ms): the update is bound by total matrix-pipe time, CUs left idle for the critical path are lost
        #   side_priority   HIP priority of the branch streams, one number or [critic, discriminator, penalty value path] (0 = default,
        #                   -1 = high; the main stream's priority is the caller's)
        #   gp_split        gp_f32 = 'x3': 'f16' = three f16 MFMAs per product on hi / lo splits of scaled operands (ASE_F32H3, ~2^-22),
        #                   'bf16' = round 4's bf16 split (ASE_F32X3, ~2^-17; penalty 1.08e-4 off in the driver's round-4 run)
        #   gp_value_late   (gp_stream off) the penalty's value path behind the loss rows' forward and heads instead of in front of them.
        #                   Measured SLOWER: 76.28 vs 75.61 ms (four interleaved repetitions, profiles/r06_schedule_options_ab.txt): off
        #   stream_offset   (measurement aid) throw-away streams taken from the pool in front of the branch streams: streams land on the
        #                   hardware queues in creation order, this shifts the engine's places
        #   gp_stream       gp_f32 modes: the penalty's value path (f32 / bf16x3 forward of the demo rows + chain: independent of
        #                   the loss rows until the conversion launch) on its own stream beside the discriminator branch.  Round 4's
        #                   A/B had it faster (with the bf16-split kernels of that round); on round 6's kernels it is SLOWER: f16gpx3
        #                   76.29 vs 74.58 ms per update without it (four interleaved repetitions on one box, a second box 77.76 vs
        #                   75.72; profiles/r06_schedule_options_ab.txt) - six more small-tile matrix launches in flight beside the
        #                   256 x 256 kernels of three streams cost more CU fragmentation than their overlap buys.  But where the
        #                   discriminator branch IS the step's critical path the own stream wins clearly: one rank's share of a sharded
        #                   step (4096 rows: 796 vs 1038 us, 2048 rows: 622 vs 919; 8192 rows: 1089 vs 1044 - off again) and under the
        #                   dynamic loss scale (95.3 vs 115.2 ms).  'auto' (default) = on for minibatches below 8192 rows and under the
        #                   dynamic loss scale, off otherwise; True / False force it
        o = dict(tn_grouped=True, tn_wg_side=64, tn_early=False, disc_early=True, short_prologue=True, style_early=False,
                 relu_bits=True, fused_apply=True, apply_wide=True, side_streams=2, gp_scale_split=True, xstep=True,
                 gp_stream='auto', style_side=0, style_wg=0, side_priority=None, prefetch=True, disc_after_style=False, gp_split='f16',
                 stream_offset=0, gp_value_late=False)
        unknown = set(cfg.get('engine_opts', {}) or {}) - set(o)
        assert not unknown, f"unknown engine_opts {sorted(unknown)}"
        o.update(cfg.get('engine_opts', {}) or {})
        self.engine_opts = o
        self._tn_defer = bool(getattr(backend, 'grouped_tn_ok', None)) and bool(o['tn_grouped'])
        self._tn_queue, self._tn_plans = [], {}          # weight gradients queued by the CURRENT branch (see _flush_tn)
        self._tn_wg_side = int(o['tn_wg_side'])
        self._tn_early = bool(o['tn_early'])
        self._disc_early = bool(o['disc_early'])
        self._early_fork = None
        # (captured hipGraphs keep the serial prologue: at config-2 size torch's capture_end segfaults on the graph of a step whose
        #  prologue is forked onto the side streams - every precision, ROCm 7.2; `profiles/r06_hipgraph_triage.txt`.  Launch programs,
        #  the default replay form and 2x faster than the captured graph anyway, are not affected)
        self._short_prologue = bool(o['short_prologue']) and cfg.get('graph_capture') != 'hipgraph'
        self._prep = self._lat_ready = self._fill_done = None
        self._style_early = bool(o['style_early'])
        self._n_side = max(1, min(int(o['side_streams']), 2))
        self._apply_groups = None
        self._use_bits = bool(o['relu_bits'])
        self._fused_apply = hasattr(backend, 'apply_multi') and bool(o['fused_apply'])
        # (a captured hipGraph forks every stream from the capturing one: an un-chained branch head cannot be captured)
        self._xstep = bool(o['xstep']) and cfg.get('graph_capture') != 'hipgraph'
        gp_side = (self.dyn_scale or self.M < 8192) if o['gp_stream'] == 'auto' else bool(o['gp_stream'])      # (self.M: THIS rank's rows)
        self._gp_side = gp_side and cfg.get('graph_capture') != 'hipgraph'      # (a fork from a forked stream: same capture_end crash)
        self._style_side = int(o['style_side'])
        self._disc_after_style = bool(o['disc_after_style'])
        self._disc_split = False
        self._enc_z_ready = False
        self._prefetch = bool(o['prefetch'])
        self._par, self._par_set, self._last_par, self._fenced = 0, False, None, False
        self._stats_exchanged = False    # this step's partial statistics were exchanged inside the un-chained heads (phase_stats)
        self._style_wg = int(o['style_wg'])
        sp = o['side_priority']
        if sp is None:
            # measured on MI355X, config 2, with the main stream high (agents: main_stream_priority): the policy's second stream
            # (critic) high and the discriminator's normal - 66.4 -> 62.9 ms (bf16); gp_f32 modes carry ~280 us more work per
            # step on the discriminator side and want that stream high instead - 73.4 -> 72.8 ms (f16gpx3)
            # (round 6: under the dynamic loss scale the ONE decision per step makes the discriminator branch the step's critical path:
            #  its stream high there too - mixed_precision 65.64 -> 64.62 ms, three interleaved repetitions, profiles/r06_schedule_options_ab.txt)
            sp = [0, -1, 0] if (self.gp32 or self.dyn_scale) else [-1, 0, 0]
        self._side_prio = [int(x) for x in sp] if isinstance(sp, (list, tuple)) else [int(sp)] * 3      # critic, disc, gp streams
        self._gp_stream_obj = None
        self._xs = False                 # this step runs the cross-step schedule (decided per step in step())
        self._disc_fwd_out = self._gp_value_done = self._pre_done = None
        self._apply_wide = bool(o['apply_wide'])
        self._apply_desc = self._apply_items = None
        self.force_dist = bool(cfg.get('force_dist', False))   # exercise the collectives with a 1-rank group
        # Gradient exchange of the data-parallel step: payload type of the two gradient buckets - 'f32' (exact: the R-rank
        # update equals the 1-rank update) or 'bf16' (half the bytes on the links: converted, summed and converted back inside the
        # exchange's host callback; the sum of R bf16-rounded partial gradients carries ~2^-9 relative error per element, which Adam's
        # normalised step tolerates and the equality tests do not - so it is an option, not the default)
        self.dp_grad_dtype = cfg.get('dp_grad_dtype', 'f32')
        assert self.dp_grad_dtype in ('f32', 'bf16'), self.dp_grad_dtype
        self._xbuf = {}
        self._disc_acc_mark = None
        self._merged_stats = False       # this step exchanges [amp sums | obs sums | mask sum] as ONE collective (phase_stats)
        self._acc_in_bucket = False      # ... and the loss partial sums inside the policy bucket's exchange (_finish_branch)
        self._refresh_desc = None
        self._mb_desc = None
        self._mb_desc_key = None
        self._build_layers()
        self._bind_params()
        self._alloc(infer_rows)
        self.refresh_shadows()

    # ------------------------------------------------------------------ structure
    def _build_layers(self):
        net, z = self.net, self.z
        act = net.activation
        split = (self.obs, P(self.obs), P(self.obs) + P(z)) if z else None
        self.style = []
        if self.kind == 'ase':
            k = z
            for i, u in enumerate(net.style_units):
                self.style.append(Dense([(f'actor_mlp._style_mlp.{2 * i}', u, 0)], k, act))
                k = u
            self.style.append(Dense([('actor_mlp._style_dense', z, 0)], k, 'tanh'))
            names_a = [f'actor_mlp._dense_layers.{i}' for i in range(len(net.units))]
            names_c = [f'critic_mlp._mlp.{2 * i}' for i in range(len(net.units))]
        else:
            names_a = [f'actor_mlp.{2 * i}' for i in range(len(net.units))]
            names_c = [f'critic_mlp.{2 * i}' for i in range(len(net.units))]
        self.actor, self.critic = [], []
        k = self.obs + z
        for i, u in enumerate(net.units):
            sp = split if (i == 0 and z) else None
            self.actor.append(Dense([(names_a[i], u, 0)], k, act, sp))
            self.critic.append(Dense([(names_c[i], u, 0)], k, act, sp))
            k = u
        self.mu_head = Dense([('mu', self.act, 0)], k, 'None')
        self.value_head = Dense([('value', 1, 0)], k, 'None')
        self.disc, self.enc_chain = [], []
        self.disc_head = self.enc_head = None
        if self.has_disc:
            k = self.amp
            for i, u in enumerate(net.disc_units):
                self.disc.append(Dense([(f'_disc_mlp.{2 * i}', u, 0)], k, net.disc_activation))
                k = u
            parts = [('_disc_logits', 1, 0)]
            if self.has_enc and not self.enc_sep:
                parts.append(('_enc', z, P(1)))
            self.disc_head = Dense(parts, k, 'None')
            if self.has_enc and self.enc_sep:
                k = self.amp
                for i, u in enumerate(net.enc_units):
                    self.enc_chain.append(Dense([(f'_enc_mlp.{2 * i}', u, 0)], k, net.enc_activation))
                    k = u
                self.enc_head = Dense([('_enc', z, 0)], k, 'None')
        self.layers = (self.style + self.actor + [self.mu_head] + self.critic + [self.value_head] + self.disc +
                       ([self.disc_head] if self.disc_head else []) + self.enc_chain +
                       ([self.enc_head] if self.enc_head else []))

    def _bind_params(self):
        net, dev, T = self.net, self.dev, self.dtype
        self.params = net.flat_params
        net._engine_bound = True          # A2CNetwork._apply refuses to re-allocate the flat buffer from now on
        n = self.params.numel()
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.n_train = net.trainable_numel           # trainable tensors come first in the flat buffer
        lr = float(self.cfg['learning_rate'])
        self.opt_state = torch.tensor([0.0, lr, 0.9, 0.999, 1e-8, 1.0, 1.0, 0.0], dtype=torch.float64, device=dev)
        # dynamic loss scale (csrc/scaler.hip): {found, skipped, growth tracker, steps, scale, growth_factor, backoff_factor,
        # growth_interval}, the table {S, 1 / S, 1 / S^2, 0} its launches read, and the optimizer state the Adam launch reads (opt_state,
        # or the identity step of a skipped step)
        self.scaler = torch.zeros(8, dtype=torch.float64, device=dev)
        self.scale_tab = torch.tensor([1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0, 0.0], dtype=torch.float32, device=dev)   # four {factor, count} records
        if self.dyn_scale:
            sc = self.loss_scaler
            self.scaler[4:8] = torch.tensor([self._gs_init, sc['growth_factor'], sc['backoff_factor'], float(int(sc['growth_interval']))],
                                            dtype=torch.float64)
            self.set_grad_scale(self._gs_init)
        self.opt_eff = self.opt_state.clone()
        self._scaler_list = None
        self._check_tabs = {}
        for d in self.layers:
            d.Ws = torch.zeros(d.n_pad, d.k_pad, dtype=T, device=dev)
            d.Wts = torch.zeros(d.k_pad, d.n_pad, dtype=T, device=dev)
            d.bs = torch.zeros(d.n_pad, dtype=torch.float32, device=dev)
            d.W, d.b, d.gW, d.gb = [], [], [], []
            for name, nr, off in d.parts:
                o, shp = net.param_slices[name + '.weight']
                assert tuple(shp) == (nr, d.K), (name, shp, nr, d.K)
                d.W.append(self.params[o:o + nr * d.K].view(nr, d.K))
                d.gW.append(self.grads[o:o + nr * d.K].view(nr, d.K))
                ob, _ = net.param_slices[name + '.bias']
                d.b.append(self.params[ob:ob + nr])
                d.gb.append(self.grads[ob:ob + nr])
        o, shp = net.param_slices['sigma']
        self.logstd = self.params[o:o + shp[0]]
        # weight-only loss terms (learning/amp_agent.py:449-466, learning/ase_agent.py:420-425): ranges of the flat buffer
        self.l2_terms = []
        c = self.cfg
        if self.has_disc:
            dc, wd, lr_ = c['disc_coef'], c['disc_weight_decay'], c['disc_logit_reg']
            for d in self.disc:
                self.l2_terms.append((d.W[0], d.gW[0], dc * 2.0 * wd, L.ACC_DISC_W2 if wd != 0 else None))
            self.l2_terms.append((self.disc_head.W[0], self.disc_head.gW[0], dc * 2.0 * (wd + lr_), None))
        if self.has_enc and c.get('enc_weight_decay', 0) != 0:
            ew = c['enc_coef'] * 2.0 * c['enc_weight_decay']
            chain = self.enc_chain if self.enc_sep else self.disc
            for d in chain:
                self.l2_terms.append((d.W[0], d.gW[0], ew, L.ACC_ENC_W2))
            eh = self.enc_head if self.enc_sep else self.disc_head
            ei = 0 if self.enc_sep else 1
            self.l2_terms.append((eh.W[ei], eh.gW[ei], ew, L.ACC_ENC_W2))

    def _alloc(self, infer_rows):
        dev, T = self.dev, self.dtype
        M, AMB = self.M, self.AMB
        Ra = 2 * M if self.div_on else M
        self.Ra = Ra

        def zt(r, c, dt=T):
            return torch.zeros(r, c, dtype=dt, device=dev)

        self._bits = {}      # activation buffer (storage pointer) -> (buffer, bit-mask twin [rows, n_pad / 32] int32)

        def with_bits(h, d):
            """Twin of an activation buffer, written by the forward epilogue (mask_out) and read by the data-gradient
            launch of the same activation: ReLU - a bit mask (1 bit instead of 2-4 bytes per element, ASE_AUX_RELU_BITS);
            smooth activations (SiLU, ELU, GELU, ...) - the pre-activation z (ASE_AUX_PREACT: act'(z) is not a function of
            the output)."""
            if d.act == L.ACT_RELU and self._use_bits:
                self._bits[h.untyped_storage().data_ptr()] = (h, torch.zeros(h.shape[0], h.shape[1] // 32, dtype=torch.int32,
                                                                                device=dev))
            elif d.act >= L.ACT_SILU:
                self._bits[h.untyped_storage().data_ptr()] = (h, torch.zeros_like(h))
            return h

        def chain_bufs(chain, rows):
            return [with_bits(zt(rows, d.n_pad), d) for d in chain], [zt(rows, d.n_pad) for d in chain]

        f32 = torch.float32
        # inputs of the first actor / critic layers ([obs | pad | latent or style code]) and the latent copies the style MLP
        # reads: TWO sets, selected by the parity of the step's result slot - the next step's prologue writes one set while the
        # previous step's weight-gradient launch still reads the other (engine_opts prefetch)
        nset = 2 if (self._prefetch and self._xstep and self.multi_stream) else 1
        self._Xa2 = [zt(Ra, self.actor[0].k_pad) for _ in range(nset)]
        self._Xc2 = [zt(M, self.critic[0].k_pad) for _ in range(nset)]
        self.Xa, self.Xc = self._Xa2[0], self._Xc2[0]
        self.Ha, self.dZa = chain_bufs(self.actor, Ra)
        self.Hc, self.dZc = chain_bufs(self.critic, M)
        self.MU = zt(Ra, self.mu_head.n_pad, f32)
        self.dMU = zt(Ra, self.mu_head.n_pad)
        self.V = zt(M, self.value_head.n_pad, f32)
        self.dV = zt(M, self.value_head.n_pad)
        if self.style:
            self._Zs2 = [zt(Ra, P(self.z)) for _ in range(nset)]
            self.Zs = self._Zs2[0]
            self.Hs, self.dZs = chain_bufs(self.style[:-1], Ra)
            self.dStyle = zt(Ra, P(self.z))
            self.new_z = zt(M, self.z, f32)
            # Philox stream of the latent draws {seed, offset}: seeded by the run, advanced on device, part of the
            # checkpoint (CommonAgent.get_full_state_weights)
            # Two streams: the ROLLOUT's latents (sample_latents(n) of the network, read-then-advance) and the in-step
            # DIVERSITY draw (advanced by begin_step, read without advancing) never share a (seed, offset) pair - with one
            # stream the first rollout draw after an update repeated the last diversity draw.  Horovod mode: rank-distinct
            # seeds (independent streams per rank, like the reference's ranks); sharded mode: one stream indexed by the
            # global row, identical on every rank.
            seed = int(self.cfg.get('seed', 0)) + (0 if self.shard else 7919 * self.rank)
            self.rng_state = torch.tensor([seed ^ 0x5EED, 0], dtype=torch.int64, device=dev)
            self.div_rng = torch.tensor([seed ^ 0xD1755EED, 0], dtype=torch.int64, device=dev)
        if self.has_disc:
            Rd = 3 * AMB
            # Rows [0, 3 AMB) = agent | replay | demo.  A 4th block of AMB rows carries the gradient-penalty chain of
            # the demo rows through the SAME launches: the chain g_l = m_l * (g_{l+1} @ W_{l+1}) is a data-gradient
            # with the demo rows' masks, and its weight-gradient terms g_l^T (dJ/dU_{l-1}) stack under dZ_l^T H_{l-1}.
            self.Xd4 = zt(4 * AMB, self.disc[0].k_pad)                   # [Xd ; s*g_0]
            self.Xd = self.Xd4[:Rd]
            self.G0 = self.Xd4[Rd:]
            self.Hd4 = [with_bits(zt(4 * AMB, d.n_pad), d) for d in self.disc]   # [H_l ; dJ/dU_l (masked)]
            self.dZd4 = [zt(4 * AMB, d.n_pad) for d in self.disc]        # [dZ_l ; s*g_l]
            self.Hd = [h[:Rd] for h in self.Hd4]
            self.dGp = [h[Rd:] for h in self.Hd4]
            self.dZd = [z[:Rd] for z in self.dZd4]
            self.Gp = [z[Rd:] for z in self.dZd4]
            self.HD = zt(Rd, self.disc_head.n_pad, f32)
            self.dHD = zt(Rd, self.disc_head.n_pad)
            # last launch of the gradient-penalty chain's backward: only its column sums are used (the penalty's gradient
            # w.r.t. the logit weights), at true scale - f32 so that half storage cannot flush them
            self.GpTop = zt(AMB, self.disc[-1].n_pad, f32)
            if self.gp32:
                assert all(d.act == L.ACT_RELU for d in self.disc), "gp_f32 is built for ReLU discriminators"
                import types
                g = self._gp32 = types.SimpleNamespace()
                g.X = zt(AMB, self.disc[0].k_pad, f32)                                  # normalised demo rows
                g.Ws = [zt(d.n_pad, d.k_pad, f32) for d in self.disc]                   # f32 shadows of the trunk
                g.Wts = [zt(d.k_pad, d.n_pad, f32) for d in self.disc]
                g.H = [zt(AMB, d.n_pad, f32) for d in self.disc]
                g.bits = [torch.zeros(AMB, d.n_pad // 32, dtype=torch.int32, device=dev) for d in self.disc]
                g.Gp = [zt(AMB, d.n_pad, f32) for d in self.disc]                       # s * g_l
                g.G0 = zt(AMB, self.disc[0].k_pad, f32)                                 # S s * g_0
                g.cast = None                                                           # (conversion table, built on first use)
            if self.enc_chain:
                self.He, self.dZe = chain_bufs(self.enc_chain, AMB)
                self.E = zt(AMB, self.enc_head.n_pad, f32)
                self.dE = zt(AMB, self.enc_head.n_pad)
            if self.has_enc:
                self.enc_z = zt(AMB, self.z, f32)     # latents of the amp rows (first amp_minibatch rows of the GLOBAL minibatch)
            if self.enc_gp:
                # encoder gradient penalty (ase_agent.py:431-441): the chain over the AGENT rows, one buffer per stage
                ech = self.enc_chain if self.enc_chain else self.disc
                ehead = self.enc_head if self.enc_chain else self.disc_head
                self.Ue = zt(AMB, ehead.n_pad)                           # s * d err / d e (the other head columns stay 0)
                self.Re = [zt(AMB, d.n_pad) for d in ech]                # s * r_l
                self.Ge = zt(AMB, ech[0].k_pad)                          # s * d err / d x
                self.Qe = [zt(AMB, d.n_pad) for d in ech]                # s * (backward of the chain), masked
                self.DUe = zt(AMB, ehead.n_pad, f32)                     # what the backward returns at u
            self.amp_mean = zt(3, self.amp, f32)
            self.amp_std = zt(3, self.amp, f32)
            self.amp_state = torch.zeros(2 * self.amp + 1, dtype=torch.float64, device=dev)
            self.amp_state[self.amp:] = 1.0
        # every per-step partial statistic that the data-parallel ranks exchange lives in ONE f64 buffer
        # [amp sums x3 | obs sums | mask sum]: one small all-reduce per step when the step runs its statistics in a
        # serial prologue; with the branch heads un-chained (cross-step schedule) each head exchanges ITS slice on its own stream -
        # the amp block from the discriminator's head, [obs | mask] (contiguous) from the policy prologue
        n_amp = 3 * 2 * self.amp if self.has_disc else 0
        self.stats_flat = torch.zeros(n_amp + 2 * self.obs + 1, dtype=torch.float64, device=dev)
        self.obs_sums = self.stats_flat[n_amp:n_amp + 2 * self.obs]
        if self.has_disc:
            self.amp_sums_flat = self.stats_flat[:n_amp]
            self.amp_sums = self.amp_sums_flat.view(3, 2 * self.amp)
        self.obs_mean = zt(1, self.obs, f32)
        self.obs_std = zt(1, self.obs, f32)
        self.obs_state = torch.zeros(2 * self.obs + 1, dtype=torch.float64, device=dev)
        self.obs_state[self.obs:] = 1.0
        self.val_state = torch.tensor([0.0, 1.0, 1.0], dtype=torch.float64, device=dev)
        self.acc = torch.zeros(L.ACC_COUNT, dtype=torch.float64, device=dev)
        self.set_result_slots(1)
        # packed f32 minibatch fields
        self.mb = {'actions': zt(M, self.act, f32), 'mu': zt(M, self.act, f32), 'sigma': zt(M, self.act, f32),
                   'old_logp_actions': zt(M, 1, f32), 'advantages': zt(M, 1, f32), 'old_values': zt(M, 1, f32),
                   'returns': zt(M, 1, f32)}
        if self.masked:
            self.mb['rand_action_mask'] = zt(M, 1, f32)
        if self.z:
            self.mb['ase_latents'] = zt(M, self.z, f32)

    # ------------------------------------------------------------------ result ring
    def set_result_slots(self, n):
        """Per-update ring of result vectors: optimisation step i of an update writes its train_result scalars into slot i
        (ase_hip_finalize_scalars' `out`) and its discriminator logits into logit slot i - nothing is snapshotted between two
        steps (round 3: two small copies per step on a side stream that had to wait for the main stream's last kernel, which
        chained the next step's discriminator head behind it).  The agent reads the whole ring once per update."""
        dev = self.dev
        self.res_ring = torch.zeros(n, L.RES_COUNT, dtype=torch.float32, device=dev)
        self.logit_ring = torch.zeros(n, 3 * self.AMB, dtype=torch.float32, device=dev) if self.has_disc else None
        self.use_slot(0)

    def use_slot(self, i):
        self.slot = i
        self.use_parity(i)
        self.res = self.res_ring[i]
        self.logit_slot = self.logit_ring[i].view(-1, 1) if self.logit_ring is not None else None

    def use_parity(self, p):
        """Input-buffer set of the next step (see _alloc: Xa / Xc / Zs exist twice under the cross-step schedule).  Consecutive
        steps must alternate - the next step's un-chained prologue writes one set while the previous step's weight-gradient
        launch still reads the other - unless the branch streams were fenced in between (fence_side_streams, which the agents
        call once per mini-epoch).  use_slot implies it; a caller that drives neither is alternated by step() itself, and a
        repeated parity without a fence is fenced there."""
        self._par = p = p % len(self._Xa2)
        self._par_set = True
        self.Xa, self.Xc = self._Xa2[p], self._Xc2[p]
        if self.style:
            self.Zs = self._Zs2[p]

    # ------------------------------------------------------------------ shadows
    def refresh_shadows(self):
        """Master f32 weights -> compute-dtype shadows (W, W^T, padded bias) for every layer: one launch."""
        if self._refresh_desc is None:
            rows, items = [], []
            for d in self.layers:
                for (name, nr, off), W, b in zip(d.parts, d.W, d.b):
                    ws, wts, bs = d.Ws[off:], d.Wts[:, off:], d.bs[off:off + nr]
                    rows.append([W.data_ptr(), nr, d.K, ws.data_ptr(), ws.stride(0), wts.data_ptr(), wts.stride(0),
                                 d.split_src, d.split_dst - d.split_src, b.data_ptr(), bs.data_ptr(), (d.K + 31) // 32])
                    items.append((W, ws, wts, d.split_src, d.split_dst, b, bs))
            self._refresh_desc = torch.tensor(rows, dtype=torch.int64, device=self.dev)
            self._refresh_items = items
        self.be.refresh_shadow_multi(self._refresh_desc, self._refresh_items, self.dtype)

    def _build_apply_desc(self):
        """Pointer table of ase_hip_apply_multi: per weight matrix its parameter / gradient / Adam-moment slices, the
        summed coefficient of its weight-only loss terms and the accumulator slots of the reported norms."""
        if self._apply_desc is not None:
            return
        import struct
        c = self.cfg
        terms = {}
        for W, gW, coef, slot in self.l2_terms:
            t = terms.setdefault(W.data_ptr(), [0.0, []])
            t[0] += coef
            if slot is not None and slot not in t[1]:
                t[1].append(slot)
        if self.has_disc:
            t = terms.setdefault(self.disc_head.W[0].data_ptr(), [0.0, []])
            t[1].insert(0, L.ACC_LOGIT_W2)
            if c['disc_weight_decay'] != 0 and L.ACC_DISC_W2 not in t[1]:
                t[1].append(L.ACC_DISC_W2)
        base = self.params.data_ptr()
        rows, items = [], []
        for d in self.layers:
            for (name, nr, off), W, b, gW, gb in zip(d.parts, d.W, d.b, d.gW, d.gb):
                ws, wts, bs = d.Ws[off:], d.Wts[:, off:], d.bs[off:off + nr]
                ow, ob = (W.data_ptr() - base) // 4, (b.data_ptr() - base) // 4
                mW, vW = self.adam_m[ow:ow + W.numel()], self.adam_v[ow:ow + W.numel()]
                mb, vb = self.adam_m[ob:ob + nr], self.adam_v[ob:ob + nr]
                coef, slots = terms.get(W.data_ptr(), [0.0, []])
                assert len(slots) <= 2, slots
                sa = slots[0] if len(slots) > 0 else -1
                sb = slots[1] if len(slots) > 1 else -1
                gap = d.split_dst - d.split_src
                # 16-byte path of the kernel: rows in whole 4-element chunks on every side (the first layers' K = obs + z
                # = 317 stays on the scalar path)
                wide = int(self._apply_wide and d.K % 4 == 0 and ws.stride(0) % 4 == 0 and wts.stride(0) % 4 == 0
                           and (d.split_src >= d.K or (d.split_src % 4 == 0 and gap % 4 == 0))
                           and ws.data_ptr() % 16 == 0 and wts.data_ptr() % 16 == 0)
                rows.append([W.data_ptr(), nr, d.K, ws.data_ptr(), ws.stride(0), wts.data_ptr(), wts.stride(0), d.split_src,
                             gap, b.data_ptr(), bs.data_ptr(), (d.K + 31) // 32, gW.data_ptr(),
                             mW.data_ptr(), vW.data_ptr(), gb.data_ptr(), mb.data_ptr(), vb.data_ptr(),
                             struct.unpack('<i', struct.pack('<f', float(coef)))[0], sa, sb, wide, 0, 0])
                items.append((W, ws, wts, d.split_src, d.split_dst, b, bs, gW.view(-1).view_as(W), mW.view_as(W), vW.view_as(W),
                              gb, mb, vb, float(coef), sa, sb))
        n_cov = sum(it[0].numel() + it[5].numel() for it in items)
        assert n_cov == self.n_train, (n_cov, self.n_train)     # every trainable scalar belongs to exactly one layer part
        self._apply_desc = torch.tensor(rows, dtype=torch.int64, device=self.dev)
        self._apply_items = items
        # parameter buckets of the two branch groups: rows of the table + the range of the flat buffers they cover
        # (checkpoint order: actor, critic, value, mu | discriminator, logits, encoder - each group is contiguous)
        n_pol = sum(len(d.parts) for d in self.style + self.actor + [self.mu_head] + self.critic + [self.value_head])

        def span(its):
            lo = min(min((it[0].data_ptr() - base) // 4, (it[5].data_ptr() - base) // 4) for it in its)
            hi = max(max((it[0].data_ptr() - base) // 4 + it[0].numel(), (it[5].data_ptr() - base) // 4 + it[5].numel())
                     for it in its)
            assert hi - lo == sum(it[0].numel() + it[5].numel() for it in its), "a parameter bucket must be contiguous"
            return lo, hi
        self._apply_groups = {'policy': (0, n_pol) + span(items[:n_pol])}
        if len(items) > n_pol:
            self._apply_groups['disc'] = (n_pol, len(items)) + span(items[n_pol:])

    # ------------------------------------------------------------------ primitive layer ops
    def _fwd(self, d, X, Y, rows, act=None):
        act = d.act if act is None else act
        self._nt(X, d.Ws, Y, rows, d.n_pad, d.k_pad, bias=d.bs, act=act,
                        mask_out=self._mask_of(Y) if (act == L.ACT_RELU or act >= L.ACT_SILU) else None)

    def _aux(self, aux, mode):
        """(aux tensor, aux mode) of a data-gradient launch: the bit-mask twin replaces a ReLU activation."""
        if mode == L.AUX_RELU_MASK:
            bits = self._mask_of(aux)
            if bits is not None:
                return bits, L.AUX_RELU_BITS
        elif (mode & 0xFF) == L.AUX_PREACT:
            pre = self._mask_of(aux)
            assert pre is not None, "a smooth activation's data gradient needs the pre-activation twin of its buffer"
            return pre, mode
        return (aux if mode else None), mode

    def _mask_of(self, h):
        """Bit-mask twin of (a row range of) an activation buffer, or None."""
        e = self._bits.get(h.untyped_storage().data_ptr()) if h is not None else None
        if e is None:
            return None
        base, bits = e
        off = (h.data_ptr() - base.data_ptr()) // base.element_size()
        r0, c0 = divmod(off, base.stride(0))
        if c0 != 0 or h.shape[1] != base.shape[1]:
            return None
        return bits[r0:r0 + h.shape[0]]

    def _twin(self, h, d):
        """Twin of activation buffer h of layer d as the gradient-penalty kernels read it (ase_hip_gp_seed / gp_second): the
        output itself for ReLU / tanh, the stored pre-activation for the smooth activations."""
        if d.act >= L.ACT_SILU:
            t = self._mask_of(h)
            assert t is not None
            return t
        return h

    def _fwd_chain(self, chain, X, H, rows):
        for d, h in zip(chain, H):
            self._fwd(d, X, h, rows)
            X = h
        return X

    def _dgrad(self, d, dY, dX, rows, aux, aux_act, wts=None, n_out=None, alpha=1.0):
        """dX = (dY @ W) * act'(aux): gradient w.r.t. the pre-activation of the layer that produced aux."""
        wts = d.Wts if wts is None else wts
        n_out = d.k_pad if n_out is None else n_out
        aux, mode = self._aux(aux, _AUX[aux_act])
        self._nt(dY, wts, dX, rows, n_out, d.n_pad, aux=aux, aux_mode=mode, alpha=alpha)

    def _wgrad(self, d, dY, X, rows, alpha=1.0, bias=True):
        """gW += dY^T X and (bias) gb += colsum(dY) from the same staged tiles.  Head groups computed their bias
        gradients in the loss-head kernels already."""
        heads = len(d.parts) > 1 or d in (self.mu_head, self.value_head, self.disc_head, self.enc_head)
        for (name, nr, off), gW, gb in zip(d.parts, d.gW, d.gb):
            self._tn(dY[:, off:], X, gW, rows, P(nr), d.k_pad, nr, d.K, d.split_src, d.split_dst, alpha=alpha,
                     gbias=gb if (bias and not heads) else None)

    def _tn(self, A, B, G, M, N, K, n_real, k_real, split_src, split_dst, alpha=1.0, gbias=None, bias_rows=0):
        """Weight (+ bias) gradient G += A^T B.  Layers the grouped kernel takes are only QUEUED here: they read buffers
        the data-gradient chain never overwrites, so phase_main launches all of them as ONE grid at its end
        (ase_hip_gemm_tn_grouped: the split-M reduction is paid once per step instead of once per layer)."""
        br = bias_rows if bias_rows > 0 else M
        alpha = alpha / self.gs              # the back-propagated operand carries the gradient scale (dynamic scale: x *_dI on the device)
        if self._tn_defer and self.be.grouped_tn_ok(A.dtype, M, n_real, K, br):
            self._tn_queue.append((A, B, G, gbias, br, M, N, K, n_real, k_real, split_src, split_dst, alpha))
        else:
            self.be.gemm_tn(A, B, G, M, N, K, n_real, k_real, split_src, split_dst, alpha=alpha, gbias=gbias,
                            bias_rows=bias_rows, alpha_dev=self._dI)

    def _flush_tn(self, target_wg=0):
        """ONE grouped launch for the weight gradients queued since the last flush (a branch of the step, or the whole step).
        target_wg: workgroups the planner sizes the work items for (0 = one per CU)."""
        q, self._tn_queue = self._tn_queue, []
        if not q:
            return
        key = (target_wg,) + tuple((a.data_ptr(), b.data_ptr(), g.data_ptr(), 0 if gb is None else gb.data_ptr(), br, M, N, K, al)
                                   for (a, b, g, gb, br, M, N, K, nr, kr, ss, sd, al) in q)
        plan = self._tn_plans.get(key)
        if plan is None:
            plan = self._tn_plans[key] = self.be.make_tn_plan(q, target_wg, alpha_dev=self._dI)
        self.be.gemm_tn_grouped(plan)

    def _bwd_chain(self, chain, X0, H, dZ, rows):
        """dZ[-1] holds d loss / d Z of the last chain layer (its bias grad already accumulated)."""
        for l in range(len(chain) - 1, -1, -1):
            d = chain[l]
            self._wgrad(d, dZ[l], H[l - 1] if l > 0 else X0, rows)
            if l > 0:
                p = chain[l - 1]
                self._dgrad(d, dZ[l], dZ[l - 1], rows, H[l - 1], p.act)

    # ------------------------------------------------------------------ one optimisation step
    def gather_minibatch(self, ds, idx, remap, part=0):
        """All small per-row fields of the minibatch (learning/amp_datasets.py:21-22) in one launch - including the two
        compute-dtype copies of the latents the first actor / critic layers read ([obs | z] without a concat).
        part 1: the fields only, part 2: the latent copies only (the short prologue launches them on different streams)."""
        key = tuple(ds[k].data_ptr() for k in self.mb)
        if self._mb_desc_key != key:
            self._mb_desc, self._mb_items, self._mb_desc_key = {}, {}, key
        part_in = part
        part = (part, self._par)           # (the latent copies go into this step's set of input buffers)
        if part not in self._mb_desc:
            rows, items = [], []
            if part_in != 2:
                for k, dst in self.mb.items():
                    src = ds[k].view(ds[k].shape[0], -1)
                    rows.append([src.data_ptr(), src.stride(0), src.shape[1], dst.data_ptr(), dst.stride(0), L.F32])
                    items.append((src, src.shape[1], dst))
            if self.z and part_in != 1:
                src = ds['ase_latents'].view(ds['ase_latents'].shape[0], -1)
                code = {torch.bfloat16: L.BF16, torch.float16: L.F16}.get(self.dtype, L.F32)
                for dst in (self.Zs[:self.M], self.Xc[:, self.actor[0].split_dst:]):
                    rows.append([src.data_ptr(), src.stride(0), self.z, dst.data_ptr(), dst.stride(0), code])
                    items.append((src, self.z, dst))
            self._mb_desc[part] = torch.tensor(rows, dtype=torch.int64, device=self.dev) if rows else None
            self._mb_items[part] = items
        if self._mb_desc[part] is not None:
            self.be.gather_multi(self._mb_desc[part], self._mb_items[part], idx, remap, self.M)

    def sync_from_rank0(self):
        """Start-up / restore: every rank takes rank 0's parameters, optimizer state, running statistics and latent
        stream position (rl_games HorovodWrapper.setup_algo: broadcast_parameters + broadcast_optimizer_state)."""
        if self.R <= 1 and not self.force_dist:
            return
        import torch.distributed as dist
        bufs = [self.params, self.adam_m, self.adam_v, self.opt_state, self.obs_state, self.val_state]
        if self.has_disc:
            bufs.append(self.amp_state)
        if self.style and self.shard:        # (Horovod mode: every rank keeps its own latent streams)
            bufs += [self.rng_state, self.div_rng]
        for t in bufs:
            if t.is_cuda and dist.get_backend() == 'gloo':
                h = t.cpu()
                dist.broadcast(h, 0)
                t.copy_(h)
            else:
                dist.broadcast(t, 0)
        self.refresh_shadows()

    def step(self, ds, idx, remap, amp_streams=None, new_z=None, apply=True, fence=True):
        """ds: dataset dict of physical-order device tensors; idx int32 [M] (this rank's rows);
        amp_streams: [(src, idx, remap)] x3 for agent / replay / demo (AMB rows each);
        new_z: optional injected diversity latents f32 [M, z] (else drawn on device).
        Phases separated by the exchange points of the data-parallel update (normaliser moments + mask sum; gradients).
        With apply (and the fused optimizer launch) every branch finishes by itself - weight gradients, gradient exchange of
        its bucket, optimizer step of its parameters - so the discriminator's tail overlaps the policy's backward.
        fence=False: the caller has ordered the branch streams behind its own writes (fence_side_streams) - required while a
        launch program is being recorded (a torch-level stream wait is not a recordable entry)."""
        inline = apply and self._fused_apply and not self.truncate
        # cross-step schedule: single GPU, streams, every branch finishing by itself (its own optimizer step)
        self._xs = bool(self._xstep and inline and self.has_disc and self._short_prologue and self._disc_early
                        and self._amp_stats_in_branch())
        if self._xs and (fence or new_z is not None):        # (injected latents are a fresh tensor of the caller's stream)
            # a caller that does not order the branch streams itself (the agents do, once per mini-epoch): whatever it did on
            # the current stream - weights loaded, statistics set, index tensors built - happens before the un-chained head
            self.fence_side_streams()
        if len(self._Xa2) == 2:
            # double-buffered inputs of the un-chained prologue (use_parity): alternate when the caller drives neither slots nor
            # parity; the same set twice in a row is only safe behind a fence of the branch streams.  While a launch program
            # records, the call RE-ISSUES the step the caller has just executed eagerly (agents._graph_step): same set, nothing
            # is launched - its replays are ordered by the alternation of the recorded slots and the caller's per-mini-epoch fences
            rec = getattr(self.be, '_recording', None) is not None
            if not rec:
                if not self._par_set:
                    self.use_parity(self._par ^ 1 if self._last_par is not None else self._par)
                if self._xs and self._par == self._last_par and not self._fenced:
                    self.fence_side_streams()
                self._last_par = self._par
            self._par_set, self._fenced = False, False
        self.phase_stats(ds, idx, remap, amp_streams, advance=apply, new_z=new_z)
        self._allreduce_stats()
        self._lr_live = apply          # (calc_gradients-style calls without the optimizer step leave the learning rate alone)
        self.phase_main(ds, idx, remap, amp_streams, new_z, inline_apply=inline)
        if inline:
            self.phase_finish()
        else:
            self._allreduce_grads()
            self.phase_apply(apply)
        return self.res

    # ---- phase A: local partial statistics -------------------------------------------------------
    def _draw_new_latents(self, new_z):
        """Latents of the diversity pass (learning/ase_agent.py:451): injected, or drawn on the device."""
        be, M = self.be, self.M
        if new_z is not None:
            be.copy_(self.new_z, new_z.contiguous())
            be.gather_rows(self.new_z, self.z, None, (0, 0), M, self.Zs[M:])
        else:
            # element index = GLOBAL minibatch row: rank r of a sharded step draws rows [r M, (r + 1) M); the kernel
            # also writes the compute-dtype copy the style MLP reads
            be.sample_latents(self.new_z, M, self.z, self.div_rng,
                              row_offset=self.rank * M if (self.shard and self.R > 1) else 0, advance=False,
                              z2=self.Zs[M:])

    def phase_stats(self, ds, idx, remap, amp_streams=None, advance=True, new_z=None):
        be, c, M, AMB = self.be, self.cfg, self.M, self.AMB
        self._disc_fwd_out = None
        self._stats_exchanged = False
        pf = self._xs and self._prefetch and len(self._Xa2) == 2
        # Sharded data parallel under the cross-step schedule: ONE statistics collective per step.  The policy prologue (critic's
        # stream) forms the observation sums and the mask sum, the discriminator's head (its own stream) the amp sums; the head waits
        # for the prologue's mark, exchanges the whole buffer [amp sums x3 | obs sums | mask sum] and the prologue waits for that -
        # every running statistic stays on the stream that owns it (round 5: three small collectives from three streams, which the
        # process group's single stream serialised anyway).
        merged = self._merged_stats = bool(pf and self._dist_shard() and self.has_disc and self.masked)
        pre_a = None
        if merged:
            self._build_apply_desc()
            with self._Branch(self, self._side(0), nowait=True):
                be.begin_step(None, None, zero2=self.stats_flat, rng_bump=self.div_rng if self.div_on else None)
                self.gather_minibatch(ds, idx, remap, part=1)            # (the fields first: the mask sum needs the gathered mask)
                be.reduce_sum(self.mb['rand_action_mask'], M, False, self.stats_flat, self.stats_flat.numel() - 1)
                if c.get('normalize_input', True):
                    be.rms_moments(ds['obs'], self.obs, idx, remap, M, self.obs_state, self.obs_sums)
                pre_a = self._mark()
        self._stats_mark_a, self._stats_mark_b = pre_a, None
        if self._xs:
            # Cross-step schedule: the head of the discriminator branch goes FIRST into the step's launch sequence and waits
            # for nothing on the main stream.  On its own stream it follows the branch's optimizer step of the previous
            # optimisation step (the only producer of what it reads: discriminator weights; the only earlier readers of what it
            # writes: that step's weight-gradient and loss-head launches, same stream) - so it runs while the main stream is
            # still in the previous step's policy tail and in this step's prologue.  It touches neither the accumulators nor
            # the optimizer state; the loss heads further down wait for begin_step.
            self._build_apply_desc()
            lo, hi = self._apply_groups['disc'][2:]
            # disc_after_style: only the HBM-bound half of the head is submitted here; its matrix launches are held back until the
            # main stream has launched the style MLP's forward (phase_main) - three narrow launches that open the step's critical
            # path and took 105 us instead of ~35 when they had to queue for CUs behind the 120-us tiles of this branch
            self._disc_split = bool(self._disc_after_style and self.style)
            with self._Branch(self, self._side(1), nowait=True):
                be.zero_(self.grads[lo:hi])
                if not merged:                   # (merged: the prologue's begin_step zeroed the whole statistics buffer)
                    be.zero_(self.amp_sums)
                if self._disc_split:
                    self._disc_inputs(amp_streams, ds)
                    self._disc_fwd_out = 'split'
                else:
                    self._disc_fwd_out = self._disc_forward(amp_streams, ds)
        pre = None
        if pf:
            # The weight-independent prologue, un-chained like the discriminator's head: on the critic's stream it follows that
            # branch's backward of the previous step (whose loss head was the last reader of the minibatch fields) and runs
            # under the previous step's policy tail.  It zeroes its own partial sums and advances the diversity stream itself
            # (one begin_step launch without optimizer state); the accumulators are not touched before the step's real
            # begin_step below (the mask sum follows it).
            with self._Branch(self, self._side(0), nowait=True) as pre:
                if merged:
                    be.wait(self._stats_mark_b)          # the one statistics collective of the step (discriminator's head) is through
                else:
                    be.begin_step(None, None, zero2=self.obs_sums, rng_bump=self.div_rng if self.div_on else None)
                if c.get('normalize_input', True):
                    if not merged:
                        be.rms_moments(ds['obs'], self.obs, idx, remap, M, self.obs_state, self.obs_sums)
                        if self._dist_shard():
                            self._ar(self.obs_sums)          # (sharded data parallel: the ranks' partial sums, on this stream)
                    be.rms_finalize(self.obs_state, self.obs, self.obs_sums, self.Mg if self.shard else self.M, 1, self.obs_mean,
                                    self.obs_std)
                else:
                    self._identity_stats(self.obs_mean, self.obs_std)
                outs = [self.Xa[:M], self.Xc]
                if self.div_on:
                    outs.append(self.Xa[M:])
                be.rms_normalize(ds['obs'], self.obs, idx, remap, M, self.obs_mean[0], self.obs_std[0], outs)
                self.gather_minibatch(ds, idx, remap, part=2)
                if self.div_on:
                    self._draw_new_latents(new_z)
                self._lat_ready = self._mark()        # everything the style MLP / the first actor layer read is in place
                if not merged:
                    self.gather_minibatch(ds, idx, remap, part=1)
        self._pre_done = pre
        if pf:
            # begin_step leaves the main stream: nothing in front of the loss heads reads the accumulators or the optimizer
            # state, so the launch (after the previous step's last kernel: mark on the main stream) and what follows it - zeroing
            # the policy's gradient bucket, the mask sum - go to the critic's stream; the main stream opens with the style MLP
            tail = self._mark()
            plo, phi = self._apply_groups['policy'][2:]
            with self._Branch(self, self._side(0), tail) as prep:           # (same stream as the prologue above: after it)
                be.begin_step(self.opt_state if advance else None, self.acc, zero2=None, rng_bump=None)
                self._early_fork = self._mark()
                be.zero_(self.grads[plo:phi])
                if merged:                               # (the global mask sum came with the statistics collective)
                    be.copy_(self.acc[L.ACC_MASK_SUM:L.ACC_MASK_SUM + 1], self.stats_flat[-1:])
                elif self.masked:
                    be.reduce_sum(self.mb['rand_action_mask'], M, False, self.acc, L.ACC_MASK_SUM)
                    if self._dist_shard():
                        self._ar(self.acc[L.ACC_MASK_SUM:L.ACC_MASK_SUM + 1])
            self._fill_done = None
            self._prep = prep
            self._stats_exchanged = True
            return
        # one launch: Adam step counter / bias corrections (advance=False - calc_gradients-style calls - leaves them),
        # loss accumulators and per-step partial statistics zeroed, position of the diversity-latent stream advanced
        be.begin_step(self.opt_state if advance else None, self.acc,
                      zero2=None if pf else (self.obs_sums if self._xs else self.stats_flat),
                      rng_bump=self.div_rng if (self.div_on and not pf) else None)
        self._prep = None
        if self._short_prologue and self._amp_stats_in_branch():
            # Short prologue (single GPU, streams): the actor chain - the critical path - keeps only the observation chain
            # (moments -> finalise -> normalise) in front of it on the main stream.  The latent copies and the diversity draw
            # (engine_opts style_early: the style MLP's three small matrix kernels too) run beside it on the critic's stream,
            # followed by the gather of the loss-head fields and the mask sum (first needed ~300 us later); zeroing the
            # gradients goes to the discriminator's stream (that branch is their first user).
            m0 = self._mark()
            with self._Branch(self, self._side(1), m0):
                be.zero_(self.grads[:self.n_train])
                self._fill_done = self._mark()
            self._early_fork = self._fill_done if self.has_disc else None
            with self._Branch(self, self._side(0), m0) as prep:
                self.gather_minibatch(ds, idx, remap, part=2)
                if self.div_on:
                    self._draw_new_latents(new_z)
                if self.style and self._style_early:
                    sd = self.actor[0].split_dst
                    h = self._fwd_chain(self.style[:-1], self.Zs, self.Hs, self.Ra)
                    self._fwd(self.style[-1], h, self.Xa[:, sd:], self.Ra)
                self._lat_ready = self._mark()
                self.gather_minibatch(ds, idx, remap, part=1)
                if self.masked:
                    be.reduce_sum(self.mb['rand_action_mask'], M, False, self.acc, L.ACC_MASK_SUM)
            self._prep = prep
            if c.get('normalize_input', True):
                be.rms_moments(ds['obs'], self.obs, idx, remap, M, self.obs_state, self.obs_sums)
            return
        be.zero_(self.grads[:self.n_train])
        # the discriminator branch needs nothing of what follows here (minibatch fields, observation moments): with its own
        # stream and no exchange between the phases it may start as soon as the accumulators and gradients are zeroed
        self._early_fork = self._mark() if (self.has_disc and self._amp_stats_in_branch() and self._disc_early) else None
        self.gather_minibatch(ds, idx, remap)
        if self.masked:
            be.reduce_sum(self.mb['rand_action_mask'], M, False, self.acc, L.ACC_MASK_SUM)
        if c.get('normalize_input', True):
            be.rms_moments(ds['obs'], self.obs, idx, remap, M, self.obs_state, self.obs_sums)
        if self.has_disc and not self._amp_stats_in_branch():
            self._amp_moments(amp_streams)

    def _amp_stats_in_branch(self):
        """With streams the amp-observation moments run at the head of the discriminator branch (its own stream) next to the
        actor / critic kernels instead of in a serial prologue.  Sharded data parallel: the head exchanges its block of partial
        sums itself (_exchange_amp_sums) - round 4 fell back to the serial prologue and, with it, to round 3's schedule whenever
        a step contained collectives.  (Captured hipGraphs cannot hold a collective: that mode keeps the serial prologue and
        the exchange between its graphs.)"""
        return self.multi_stream and not (self._dist_shard() and self.cfg.get('graph_capture') == 'hipgraph')

    def _dist_shard(self):
        return self._dist_on() and self.shard

    def _exchange_amp_sums(self):
        if self._merged_stats:
            # the step's ONE statistics collective: behind the policy prologue's sums (mark A), in front of everything that reads a
            # global statistic on either stream (mark B)
            self.be.wait(self._stats_mark_a)
            self._ar(self.stats_flat)
            self._stats_mark_b = self._mark()
        elif self._dist_shard():
            self._ar(self.amp_sums_flat)

    def _amp_moments(self, amp_streams):
        if self.cfg.get('normalize_amp_input', True):       # (the partial sums were zeroed by begin_step)
            self.be.rms_moments_multi(amp_streams, self.amp, self.AMB, self.amp_state, [self.amp_sums[s] for s in range(3)])

    # ---- fork / join of the independent actor / critic / discriminator branches -------------------
    def _side(self, k):
        """k-th side stream (HIP backend only; the branches are independent until the loss heads / the optimizer,
        and the small tail-heavy kernels of one branch fill the CUs another leaves idle)."""
        if not self.multi_stream:
            return None
        if self._side_streams is None:
            self._pad_streams = [torch.cuda.Stream(device=self.dev, priority=p) for _ in range(int(self.engine_opts['stream_offset']))
                                 for p in (0, -1)]
            self._side_streams = [torch.cuda.Stream(device=self.dev, priority=self._side_prio[k % 3]) for k in range(self._n_side)]
            if self.gp32 and not self._gp_side and self._gp_stream_obj is None:
                # A gp_f32 engine takes its value-path stream from the pool even when it does not use it (gp_stream off): streams
                # land on the hardware queues in creation order, and an engine that takes THREE streams shifts every later engine of
                # the process by one place - the benchmark's later legs (bf16 throughput mode, the 16384-environment batch) then ran
                # with two of their branch streams on one hardware queue: 82-85 ms instead of 62, 371-374 instead of 296-303
                # (profiles/r06_bench_n1.json's history: calls AC / AE against I / M)
                self._gp_stream_obj = torch.cuda.Stream(device=self.dev, priority=self._side_prio[2])
        return self._side_streams[k % len(self._side_streams)]

    def fence_side_streams(self):
        """Everything the main stream holds so far happens before whatever the branch streams are given next.  Inside a step
        the branches fork from marks on the main stream - except the head of the discriminator branch under the cross-step
        schedule (engine_opts xstep), which waits for nothing: the agent calls this after it has written what that head reads
        (per-mini-epoch index buffers, the demo / replay rings), once per mini-epoch."""
        if not self.multi_stream or self.dev.type != 'cuda':
            return
        cur = torch.cuda.current_stream(self.dev)
        self._side(0)                                     # (streams exist before their first use: a fresh stream is unordered)
        for st in list(self._side_streams) + ([self._gp_stream()] if self.gp32 and self._gp_side else []):
            st.wait_stream(cur)
        self._fenced = True

    def fence_main_behind_sides(self, main):
        """`main` waits for everything the branch streams hold (error path of the agents' update(): inside a step the branches
        are joined by marks, which an exception may have skipped)."""
        if not self.multi_stream or self.dev.type != 'cuda' or self._side_streams is None:
            return
        for st in list(self._side_streams) + ([self._gp_stream_obj] if self._gp_stream_obj is not None else []):
            main.wait_stream(st)

    class _Branch:
        """Run a block of launches on a side stream: it starts after `after` (a mark on the main stream; default: everything
        the main stream holds so far) and leaves `done` for whoever needs its results.  Marks / waits go through the
        backend (ase_hip_mark / ase_hip_wait), so a recorded launch program contains them."""

        def __init__(self, eng, stream, after=None, nowait=False):
            self.eng, self.stream, self.after, self.nowait = eng, stream, after, nowait

        def __enter__(self):
            if self.stream is not None:
                after = None if self.nowait else (self.after if self.after is not None else self.eng.be.mark())
                self.ctx = torch.cuda.stream(self.stream)
                self.ctx.__enter__()
                if after is not None:
                    self.eng.be.wait(after)
            return self

        def __exit__(self, *a):
            if self.stream is not None:
                self.done = self.eng.be.mark()
                self.ctx.__exit__(*a)

    def _join_branch(self, br):
        if br.stream is not None:
            self.be.wait(br.done)

    def _mark(self):
        """Mark at the current position of the main stream (fork point of later branches), None without streams."""
        return self.be.mark() if self.multi_stream else None

    def _finish_branch(self, group, inline_apply, last=False):
        """Tail of a branch: its weight gradients as one grouped launch, then (inline_apply) the exchange of its gradient
        bucket and the optimizer step of its parameters.  Side branches size their grouped launch for part of the chip
        (they run beside other branches' kernels; every work item pays 256 KB of atomics, so fewer, longer items)."""
        self._flush_tn(0 if last else self._tn_wg_side)
        if inline_apply:
            a, b, lo, hi = self._apply_groups[group]
            if self._dist_on():
                # the LAST bucket of a sharded step (the policy's, on the main stream) carries the loss partial sums of the whole step
                # with it: one collective less (round 5: a 100-byte all-reduce of its own in phase_finish).  The discriminator branch's
                # contributions to them are complete at its mark (_disc_acc_mark), long before this point.
                with_acc = bool(last and self.shard and self.dp_grad_dtype == 'f32')
                if with_acc and self._disc_acc_mark is not None:
                    self.be.wait(self._disc_acc_mark)
                self._exchange_bucket(group, lo, hi, with_acc)
                self._acc_in_bucket = with_acc
                if not self.shard:
                    self._host(lambda: self.grads[lo:hi].mul_(1.0 / self.R))
            if self.dyn_scale:
                # GradScaler's decision is ONE per step (learning/ase_agent.py:271-288: one optimizer over every parameter): the
                # branch checks what its launches could not report themselves and its (exchanged) f32 gradient, here on its own
                # stream; the optimizer step of BOTH buckets follows the last branch (_dyn_apply)
                self._dyn_check(group, lo, hi)
                if last:
                    self._dyn_apply()
            else:
                self.be.apply_multi(self._apply_desc[a:b], self._apply_items[a:b], self.dtype, self.opt_state, self.acc)

    def _disc_forward(self, amp_streams, ds=None):
        """Head of the discriminator (+ encoder) branch: AMP-observation moments -> running statistics -> normalised rows
        [agent | replay | demo] -> trunk forward -> joint [logit | enc] head (+ the separate encoder's chain).  Needs the
        branch's weights and nothing of the step's accumulators."""
        self._disc_inputs(amp_streams, ds)
        return self._disc_matrices()

    def _disc_inputs(self, amp_streams, ds=None):
        """First half of the branch's head: statistics and normalised inputs (HBM-bound streams) + the fork of the penalty's
        value path (gp_f32 modes)."""
        be, c, AMB = self.be, self.cfg, self.AMB
        Rd = 3 * AMB
        norm_amp = self.has_disc and c.get('normalize_amp_input', True)
        amb_den = self.AMBg if self.shard else self.AMB
        if self._amp_stats_in_branch():
            self._amp_moments(amp_streams)
            self._exchange_amp_sums()
        if norm_amp:
            be.rms_finalize(self.amp_state, self.amp, self.amp_sums, amb_den, 3, self.amp_mean, self.amp_std)
        else:
            self._identity_stats(self.amp_mean, self.amp_std)
        gp_fork = self._mark() if (self.gp32 and self._gp_side) else None
        self._enc_z_ready = False
        if self.has_enc and ds is not None:
            # enc_latents = ase_latents[0:amp_minibatch] (learning/ase_agent.py:247): a row gather that needs nothing but the
            # step's indices - with the un-chained head instead of between the discriminator's loss heads
            src, sidx, srm = amp_streams[0]
            zsrc = ds['ase_latents'].view(ds['ase_latents'].shape[0], -1)
            be.gather_rows(zsrc, self.z, sidx, srm, AMB, self.enc_z)
            self._enc_z_ready = True
        xd = [self.Xd[s * AMB:(s + 1) * AMB] for s in range(3)]
        if self.amp % 4 == 0 and all(src.stride(0) % 4 == 0 for src, _, _ in amp_streams):
            be.rms_normalize_multi(amp_streams, self.amp, AMB, [self.amp_mean[s] for s in range(3)],
                                   [self.amp_std[s] for s in range(3)], xd)
        else:                          # rows that are not whole 16-byte chunks: one launch per stream
            for s, (src, sidx, srm) in enumerate(amp_streams):
                be.rms_normalize(src, self.amp, sidx, srm, AMB, self.amp_mean[s], self.amp_std[s], [xd[s]])
        self._gp_value_done = None
        if self.gp32:
            gp_coef = c['disc_coef'] * c['disc_grad_penalty']
            if gp_fork is not None:
                # the penalty's value path beside the loss rows' forward, on its own stream: it follows the statistics above
                # (and, through them, the branch's previous optimizer step) and is joined before the conversion launch
                with self._Branch(self, self._gp_stream(), gp_fork) as br:
                    self._gp_value(amp_streams, gp_coef)
                self._gp_value_done = br
            elif self.engine_opts['gp_value_late']:
                self._gp_value_pending = (amp_streams, gp_coef)         # launched by _gp_f32, behind the loss rows' forward and heads
            else:
                self._gp_value(amp_streams, gp_coef)

    def _disc_matrices(self):
        """Second half: the matrix launches (trunk forward, joint head) and the logit snapshot."""
        be, AMB = self.be, self.AMB
        Rd = 3 * AMB
        hd = self._fwd_chain(self.disc, self.Xd, self.Hd, Rd)
        self._fwd(self.disc_head, hd, self.HD, Rd)
        if self.logit_slot is not None:        # the step's logits into its slot of the result ring (train_result's disc_*_logit)
            be.gather_rows(self.HD, 1, None, (0, 0), Rd, self.logit_slot)
        he = None
        if self.enc_chain:
            he = self._fwd_chain(self.enc_chain, self.Xd[:AMB], self.He, AMB)
            self._fwd(self.enc_head, he, self.E, AMB)
        return hd, he

    def _gp_stream(self):
        if not self.multi_stream:
            return None
        if self._gp_stream_obj is None:
            self._gp_stream_obj = torch.cuda.Stream(device=self.dev, priority=self._side_prio[2])
        return self._gp_stream_obj

    # ---- phase B: normalise, forward, loss heads, backward -----------------------------------------
    def phase_main(self, ds, idx, remap, amp_streams=None, new_z=None, inline_apply=False):
        be, c, M, AMB = self.be, self.cfg, self.M, self.AMB
        norm_in = c.get('normalize_input', True)
        norm_amp = self.has_disc and c.get('normalize_amp_input', True)
        Ra = self.Ra
        if inline_apply:
            self._build_apply_desc()
        fork0 = self._early_fork if self._early_fork is not None else self._mark()   # (nothing of the observation prologue)
        disc_early = self.has_disc and self._early_fork is not None and self._disc_early
        amb_den = self.AMBg if self.shard else self.AMB
        Rd = 3 * AMB

        if self._disc_fwd_out == 'split':            # (its matrix half follows the style forward below)
            hd = he = None
        elif self._disc_fwd_out is not None:         # cross-step schedule: submitted at the top of phase_stats
            hd, he = self._disc_fwd_out
        elif disc_early:
            with self._Branch(self, self._side(1), fork0):
                hd, he = self._disc_forward(amp_streams)
        if self._pre_done is None:            # (prefetch: statistics + normalised inputs came with the un-chained prologue)
            if norm_in:
                be.rms_finalize(self.obs_state, self.obs, self.obs_sums, self.Mg if self.shard else self.M, 1, self.obs_mean,
                                self.obs_std)
            else:
                self._identity_stats(self.obs_mean, self.obs_std)
            outs = [self.Xa[:M], self.Xc]
            if self.div_on:
                outs.append(self.Xa[M:])
            be.rms_normalize(ds['obs'], self.obs, idx, remap, M, self.obs_mean[0], self.obs_std[0], outs)
        fork1 = self._mark()                 # critic: observations normalised, latents in place (gather_minibatch)
        if self._prep is not None:
            be.wait(self._lat_ready)         # latent copies + diversity draw of the short prologue (critic's stream)
        elif self.div_on:
            self._draw_new_latents(new_z)

        # The actor chain (2 M rows with the diversity pass) is the longest: it is launched FIRST on the main stream, the
        # critic and the discriminator branches follow on their streams, forked from the events above.
        # (style_early ran the style MLP with the short prologue on the critic's stream; the un-chained prologue of `prefetch`
        #  cannot - it precedes the optimizer step that writes the style weights - so the option is ignored there)
        if self.style and not (self._prep is not None and self._style_early and self._pre_done is None):
            sd = self.actor[0].split_dst
            h = self._fwd_chain(self.style[:-1], self.Zs, self.Hs, Ra)
            self._fwd(self.style[-1], h, self.Xa[:, sd:], Ra)
        style_launched = self._mark() if self._disc_fwd_out == 'split' else None
        ha = self._fwd_chain(self.actor, self.Xa, self.Ha, Ra)
        self._fwd(self.mu_head, ha, self.MU, Ra)

        with self._Branch(self, self._side(0), fork1) as br_critic:
            hc = self._fwd_chain(self.critic, self.Xc, self.Hc, M)
            self._fwd(self.value_head, hc, self.V, M)

        # -- discriminator (+ encoder) branch: normalise, forward, heads, backward, gradient penalty, its optimizer step
        br_disc = None
        if self.has_disc:
            tnq, self._tn_queue = self._tn_queue, []          # the branch queues (and flushes) its own weight gradients
            with self._Branch(self, self._side(1), fork0) as br_disc:
                if self._disc_fwd_out == 'split':
                    be.wait(style_launched)
                    hd, he = self._disc_matrices()
                elif not disc_early:
                    hd, he = self._disc_forward(amp_streams)
                be.disc_head(self.HD, self.dHD, self.disc_head.gb[0], self.acc, AMB, amb_den, c['disc_coef'],
                             grad_scale=self.gs, dyn=self._dS)
                if self.has_enc:
                    src, sidx, srm = amp_streams[0]   # enc_latents = ase_latents[0:amp_minibatch] (learning/ase_agent.py:247)
                    zsrc = ds['ase_latents'].view(ds['ase_latents'].shape[0], -1)
                    if not self._enc_z_ready:
                        be.gather_rows(zsrc, self.z, sidx, srm, AMB, self.enc_z)
                    if self.enc_sep:
                        be.enc_head(self.E, self.enc_z, self.dE, self.enc_head.gb[0], None, self.acc, AMB, amb_den,
                                    self.z, c['enc_coef'], grad_scale=self.gs, dyn=self._dS)
                    else:
                        off = self.disc_head.parts[1][2]
                        be.enc_head(self.HD[:AMB, off:], self.enc_z, self.dHD[:AMB, off:], self.disc_head.gb[1], None,
                                    self.acc, AMB, amb_den, self.z, c['enc_coef'], grad_scale=self.gs, dyn=self._dS)
                if self.enc_gp:
                    self._enc_grad_penalty(he if self.enc_chain else hd[:AMB])
                self._wgrad(self.disc_head, self.dHD, hd, Rd)
                last = self.disc[-1]
                self._dgrad(self.disc_head, self.dHD, self.dZd[-1], Rd, self.Hd[-1], last.act)
                self._disc_backward()
                if self.enc_chain:
                    self._wgrad(self.enc_head, self.dE, he, AMB)
                    last = self.enc_chain[-1]
                    self._dgrad(self.enc_head, self.dE, self.dZe[-1], AMB, self.He[-1], last.act)
                    self._bwd_chain(self.enc_chain, self.Xd[:AMB], self.He, self.dZe, AMB)
                # (every contribution of this branch to the loss partial sums - logit losses, penalties, encoder loss - is launched)
                self._disc_acc_mark = self._mark()
                self._finish_branch('disc', inline_apply)
            self._tn_queue = tnq
        self._join_branch(br_critic)
        if self._prep is not None:
            self._join_branch(self._prep)      # (same stream as the critic branch: already implied; kept explicit)
            if self._fill_done is not None:
                be.wait(self._fill_done)       # gradients zeroed (discriminator's stream) before the loss head adds to them

        # -- PPO loss head (value + gradient w.r.t. mu / value + head bias gradients)
        be.ppo_head(self.MU, self.V, self.mb, self.new_z if self.div_on else None, self.logstd, self.dMU, self.dV,
                    self.mu_head.gb[0], self.value_head.gb[0], self.acc, M, self.Mg if self.shard else self.M, self.act, self.z,
                    self.masked, self.div_on, self.mu_tanh, c['clip_value'], c['e_clip'], c['critic_coef'],
                    self.bounds_coef, c.get('amp_diversity_bonus', 0.0), c.get('amp_diversity_tar', 0.0), grad_scale=self.gs, dyn=self._dS)
        fork2 = self._mark()

        # -- actor backward on the main stream, critic backward beside it.  The wide layers' weight gradients of BOTH
        # (one parameter bucket) go out as one grouped launch on the critic's stream as soon as the actor's data-gradient
        # chain is through, next to the small kernels of the style-MLP backward that end the main stream's chain.
        self._wgrad(self.mu_head, self.dMU, ha, Ra)
        last = self.actor[-1]
        self._dgrad(self.mu_head, self.dMU, self.dZa[-1], Ra, self.Ha[-1], last.act)
        self._bwd_chain(self.actor, self.Xa, self.Ha, self.dZa, Ra)
        tn_actor, self._tn_queue = self._tn_queue, []
        actor_done = self._mark()
        with self._Branch(self, self._side(0), fork2) as br_cb:
            self._wgrad(self.value_head, self.dV, hc, M)
            last = self.critic[-1]
            self._dgrad(self.value_head, self.dV, self.dZc[-1], M, self.Hc[-1], last.act)
            self._bwd_chain(self.critic, self.Xc, self.Hc, self.dZc, M)
            if self._tn_early:
                if actor_done is not None:
                    be.wait(actor_done)
                self._tn_queue = tn_actor + self._tn_queue
                self._flush_tn(0)
            else:
                tn_actor = tn_actor + self._tn_queue
                self._tn_queue = []
        def style_backward():
            a0, sdn = self.actor[0], self.style[-1]
            sd = a0.split_dst
            self._dgrad(a0, self.dZa[0], self.dStyle, Ra, self.Xa[:, sd:], sdn.act, wts=a0.Wts[sd:], n_out=P(self.z))
            hs_last = self.Hs[-1] if self.Hs else self.Zs
            self._wgrad(sdn, self.dStyle, hs_last, Ra)
            if self.Hs:
                p = self.style[-2]
                self._dgrad(sdn, self.dStyle, self.dZs[-1], Ra, self.Hs[-1], p.act)
                self._bwd_chain(self.style[:-1], self.Zs, self.Hs, self.dZs, Ra)

        if self.style and self._style_side == 2 and self.multi_stream and self._tn_defer and not self._tn_early:
            # variant: only the style MLP's three data-gradient launches leave the critical path (critic's stream, beside the
            # wide weight-gradient launch); its weight gradients follow the wide launch on the main stream as a second grouped
            # launch over the whole chip
            self._join_branch(br_cb)
            wide, self._tn_queue = tn_actor, []
            with self._Branch(self, self._side(0), actor_done) as br_style:
                style_backward()
            narrow, self._tn_queue = self._tn_queue, wide
            self._flush_tn(0)
            self._join_branch(br_style)
            self._tn_queue = narrow
        elif self.style and self._style_side and self.multi_stream and self._tn_defer and not self._tn_early:
            # The style MLP's backward (three narrow data-gradient launches, ~75 us back to back) used to sit between the actor's
            # data-gradient chain and the policy's grouped weight-gradient launch - on the step's critical path.  The wide
            # launch needs nothing of it: it goes out as soon as the actor's and the critic's chains are through, and the style
            # backward runs beside it on the critic's stream (idle by then) with its OWN small grouped launch for the three
            # style weight gradients (sized for the CUs the wide launch leaves free); the optimizer step waits for both.
            self._join_branch(br_cb)
            wide, self._tn_queue = tn_actor, []
            with self._Branch(self, self._side(0), actor_done) as br_style:
                if self._style_wg <= 0:          # style weight gradients as direct launches of the 128 x 128 split-M kernel
                    defer, self._tn_defer = self._tn_defer, False
                    style_backward()
                    self._tn_defer = defer
                else:
                    style_backward()
                    self._flush_tn(self._style_wg)
            self._tn_queue = wide
            self._flush_tn(0)
            self._join_branch(br_style)
        else:
            if self.style:
                style_backward()
            self._join_branch(br_cb)
            if not self._tn_early:
                self._tn_queue = tn_actor + self._tn_queue
        self._finish_branch('policy', inline_apply, last=True)        # (flushes what is still queued)
        if br_disc is not None:
            self._join_branch(br_disc)

    def phase_finish(self):
        """After the inline per-branch optimizer steps: loss partial sums over the ranks (weight-norm slots are local),
        reported scalars."""
        c = self.cfg
        if self._dist_on() and self.shard and not self._acc_in_bucket:
            self._ar(self.acc[1:L.ACC_LOGIT_W2])
        self._acc_in_bucket = False
        self._average_kl()
        self.be.finalize_scalars(self.acc, self.res, self.Mg if self.shard else self.M, self.AMBg if self.shard else self.AMB,
                                 self.masked, self.has_disc, self.has_enc, self.div_on, c,
                                 opt_state=self.opt_state if (self.adaptive_lr and self._lr_live) else None, kl_threshold=self.kl_threshold)

    # ---- phase C (end-of-step form): weight-only loss terms, optimizer, shadows, reported scalars ------
    def phase_apply(self, apply=True):
        be, c = self.be, self.cfg
        if apply and self._fused_apply and not self.truncate and not self.dyn_scale:      # (dynamic scale + fused: step() took the inline form)
            # weight-only loss terms + their reported norms + Adam + shadow refresh of every layer: ONE launch
            self._build_apply_desc()
            be.apply_multi(self._apply_desc, self._apply_items, self.dtype, self.opt_state, self.acc)
        else:
            if self.has_disc:
                # weight-only loss terms, added once after the gradient reduction
                for W, gW, coef, slot in self.l2_terms:
                    if coef != 0:
                        be.axpy(gW.view(-1), W.view(-1), coef)
                wl = self.disc_head.W[0]
                be.reduce_sum(wl.view(-1), wl.numel(), True, self.acc, L.ACC_LOGIT_W2)
                if c['disc_weight_decay'] != 0:
                    for d in self.disc:
                        be.reduce_sum(d.W[0].view(-1), d.W[0].numel(), True, self.acc, L.ACC_DISC_W2)
                    be.reduce_sum(wl.view(-1), wl.numel(), True, self.acc, L.ACC_DISC_W2)
                if self.has_enc and c.get('enc_weight_decay', 0) != 0:
                    for W, gW, coef, slot in self.l2_terms:
                        if slot == L.ACC_ENC_W2:
                            be.reduce_sum(W.view(-1), W.numel(), True, self.acc, L.ACC_ENC_W2)
            if apply and self.dyn_scale:
                # GradScaler (learning/ase_agent.py:271-288): found_inf over the scaled backward (every half buffer a launch of the
                # step wrote + the f32 gradient), one flag for all ranks, then the decision - a found overflow zeroes the gradient
                # and the optimizer launch below runs the identity step (weights, moments, step counter stay what they are)
                g = self.grads[:self.n_train]
                self._dyn_check('all', 0, self.n_train)
                if self._dist_on():
                    be.scaler_fold(self.scaler, self.scale_tab)
                    self._ar(self.scaler[:1])
                be.scaler_step(self.scaler, self.opt_state, self.opt_eff, g, scale_tab=self.scale_tab)
            elif self.dyn_scale:
                # a step without the optimizer (calc_gradients-style calls): GradScaler sees nothing of it - what its launches
                # reported is dropped
                self._host(lambda: self.scale_tab[1::2].zero_())
            if apply and self.truncate:
                g = self.grads[:self.n_train]
                be.reduce_sum(g, g.numel(), True, self.acc, L.ACC_GRAD_SQ)
                be.clip_scale(g, self.acc, L.ACC_GRAD_SQ, self.grad_norm)
            if apply:
                be.adam(self.params[:self.n_train], self.grads[:self.n_train], self.adam_m[:self.n_train],
                        self.adam_v[:self.n_train], self.opt_eff if self.dyn_scale else self.opt_state)
                self.refresh_shadows()
        self._average_kl()
        be.finalize_scalars(self.acc, self.res, self.Mg if self.shard else self.M, self.AMBg if self.shard else self.AMB,
                            self.masked, self.has_disc, self.has_enc, self.div_on, c,
                            opt_state=self.opt_state if (self.adaptive_lr and apply) else None, kl_threshold=self.kl_threshold)

    # ---- dynamic loss scale (cfg loss_scale = 'dynamic') ---------------------------------------------------------------
    def _scaler_bufs(self):
        """Buffers of the step that half conversions write and whose writers carry NO scale record, by branch - the few the
        producers' own reports (ABI 7: every NT matrix launch and every loss head of a dynamic-scale step is given a record, _nt)
        leave over: seeds and second-order terms of the penalty chains (ase_hip_gp_seed / gp_second / enc_gp_seed) and the one
        conversion launch of a gp_f32 engine.  Rounds 4-5 listed every half buffer here and re-read 0.82 GB per step.
        Built from the engine's structure: a buffer a feature allocates MUST be found (a renamed attribute is an assertion, not a
        silently shorter list)."""
        if self._scaler_list is None:
            out = []

            def add(x):
                for t in (x if isinstance(x, (list, tuple)) else [x]):
                    assert t is not None
                    if t.numel():
                        assert t.is_contiguous()
                        out.append(t)
            c = self.cfg
            need = []
            gp_on = self.has_disc and c['disc_coef'] * c['disc_grad_penalty'] != 0
            if gp_on and self.gp32:
                need += ['Gp', 'G0']                      # the conversion launch's outputs (16-bit copies of the exact chain)
            elif gp_on:
                need += ['Gp']                            # gp_seed writes the top of the chain (the rest: matrix launches)
                if any(d.act != L.ACT_RELU for d in self.disc):
                    need += ['dZd4']                      # gp_second adds the second-order terms into the demo rows
            if self.enc_gp:
                need += ['Ue']                            # enc_gp_seed
            for name in need:
                assert hasattr(self, name), f"_scaler_bufs: the engine has no buffer '{name}' (renamed?)"
                add(getattr(self, name))
            if gp_on and self.gp32:
                add(self._gp32.Gp[-1])                    # gp_seed's f32 output
            self._scaler_list = {'disc': out, 'policy': []}
        return self._scaler_list

    def _dyn_check(self, group, lo, hi):
        """found_inf over what the launches of a branch ('disc' / 'policy'; 'all': the end-of-step form) did not report themselves
        + the branch's f32 gradient bucket [lo, hi) (after its exchange: the same bits on every rank): ONE launch."""
        bufs = self._scaler_bufs()
        key = (group, lo, hi)
        tab = self._check_tabs.get(key)
        if tab is None:
            lst = (bufs['disc'] + bufs['policy']) if group == 'all' else list(bufs[group])
            lst.append(self.grads[lo:hi])
            tab = self._check_tabs[key] = (lst, self.be.make_check_table(lst) if hasattr(self.be, 'make_check_table') else None)
        self.be.scaler_check_multi(tab[0], self.scaler, table=tab[1])

    def _dyn_apply(self):
        """GradScaler.step + update + the optimizer step of EVERY parameter, after the last branch's checks.  With streams it is
        submitted on the discriminator branch's stream - behind that branch's tail, waiting for the main stream's position - so
        that the head of the NEXT step's discriminator branch (cross-step schedule: un-chained, same stream) follows the
        optimizer step by stream order; the main stream joins it."""
        be = self.be
        g = self.grads[:self.n_train]
        side = self._side(1) if (self.multi_stream and self.has_disc) else None
        with self._Branch(self, side) as br:
            if self._dist_on():
                be.scaler_fold(self.scaler, self.scale_tab)       # the producers' reports -> one number, SUM over the ranks
                self._ar(self.scaler[:1])
            be.scaler_step(self.scaler, self.opt_state, self.opt_eff, g, scale_tab=self.scale_tab)
            be.apply_multi(self._apply_desc, self._apply_items, self.dtype, self.opt_eff, self.acc)
        self._join_branch(br)

    @property
    def _dS(self):
        """Scale records of the dynamic loss scale for the `*_dev` arguments ({factor, overflow count}, include/ase_hip.h): S (loss
        heads), 1 / S (_dI: weight gradients, the top of the penalty chain), 1 / S^2 (_dI2: the penalty's norm), 1 (_dOne: matrix
        launches without a factor of their own - they still REPORT what they store); None under the static scale, where self.gs
        carries it."""
        return self.scale_tab[0:2] if self.dyn_scale else None

    @property
    def _dI(self):
        return self.scale_tab[2:4] if self.dyn_scale else None

    @property
    def _dI2(self):
        return self.scale_tab[4:6] if self.dyn_scale else None

    @property
    def _dOne(self):
        return self.scale_tab[6:8] if self.dyn_scale else None

    def _nt(self, *a, alpha_dev=None, **kw):
        """Every NT matrix launch of the step: under the dynamic loss scale it carries a scale record - the factor it was given, or
        the record of factor 1 - so that the launch itself reports an overflow it stores (GradScaler's found_inf without a pass over
        the step's buffers; rounds 4-5 re-read 0.82 GB per step for it)."""
        if alpha_dev is None and self.dyn_scale:
            alpha_dev = self._dOne
        self.be.gemm_nt(*a, alpha_dev=alpha_dev, **kw)

    def set_grad_scale(self, s):
        """A new gradient scale (a power of two).  Static scale: a launch ARGUMENT of the loss heads and the weight-gradient launches -
        recorded launch programs / captured graphs of the step hold the old one and must be dropped by their owner.  Dynamic scale:
        written into the device state (GradScaler's `_scale`) and its table; recorded programs read it there and stay valid."""
        assert s > 0 and math.log2(s) == round(math.log2(s)), s
        if self.dyn_scale:
            s = float(s)
            self.scaler[4:5] = torch.tensor([s], dtype=torch.float64)
            self.scale_tab.copy_(torch.tensor([s, 0.0, 1.0 / s, 0.0, 1.0 / (s * s), 0.0, 1.0, 0.0], dtype=torch.float32))
        else:
            self.gs = float(s)

    def scaler_state(self):
        """{'scale', 'skipped', 'clean' (the growth tracker), 'steps'} - reads the device state (synchronises)."""
        f, sk, cl, st, sc = self.scaler[:5].tolist()
        return {'scale': sc if self.dyn_scale else self.gs, 'skipped': int(sk), 'clean': int(cl), 'steps': int(st)}

    def scaler_update(self):
        """GradScaler.update() (torch/amp/grad_scaler.py) happens on the DEVICE after every optimisation step (ase_hip_scaler_step:
        scale *= backoff_factor after a step with found_inf, *= growth_factor after growth_interval clean steps in a row) - as the
        reference calls scaler.update() behind every scaler.step() (learning/ase_agent.py:280,285,288).  Nothing is left for the host
        to do between updates and no recorded program goes stale: kept for callers of the round-5 interface, returns False."""
        assert self.dyn_scale
        return False

    def _enc_grad_penalty(self, h_top):
        """Encoder gradient penalty  c * mean_rows |d err / d x|^2,  err = -<normalize(e), z>  (learning/ase_agent.py:431-441,
        a double backward in the reference), x = the normalised AMP observation of the agent rows, c = enc_coef *
        enc_grad_penalty.  With ReLU layers h_l = relu(W_l h_{l-1} + b_l), e = W_e h_L + b_e and masks m_l = [h_l > 0]:
            u = d err / d e                       (per row: ase_hip_enc_gp_seed)
            r_L = m_L * (W_e^T u),  r_{l-1} = m_{l-1} * (W_l^T r_l),  g = W_1^T r_1 = d err / d x     (data-gradient launches)
            penalty = sum |g|^2 / AMB
        and its gradient, with dg = (2 c / AMB) g  (masks are constants almost everywhere):
            q_1 = m_1 * (W_1 dg),  q_l = m_l * (W_l q_{l-1}),  du = W_e q_L                          (forward-shaped launches)
            gW_1 += r_1^T dg,  gW_l += r_l^T q_{l-1},  gW_e += u^T q_L                               (weight-gradient launches)
            d_e += (d u / d e) du          (per row: ase_hip_enc_gp_back; from there the ordinary backward carries it,
                                            including the bias gradients)
        Everything is carried scaled by s = sqrt(2 c / AMB) so that the weight-gradient launches need alpha = 1.  Runs on the
        agent rows only (AMB): with the shared trunk the layers are the discriminator's and u sits in the encoder columns of
        the joint head (logit column 0), with enc.separate the encoder's own chain."""
        be, c, AMB = self.be, self.cfg, self.AMB
        sep = bool(self.enc_chain)
        chain = self.enc_chain if sep else self.disc
        head = self.enc_head if sep else self.disc_head
        H = self.He if sep else [h[:AMB] for h in self.Hd]
        X0 = self.Xd[:AMB]
        for d in chain:
            assert d.act == L.ACT_RELU, "analytic gradient penalty needs ReLU encoder layers"
        if sep:
            e, d_e, db, off = self.E, self.dE, self.enc_head.gb[0], 0
        else:
            off = self.disc_head.parts[1][2]
            e, d_e, db = self.HD[:AMB, off:], self.dHD[:AMB, off:], self.disc_head.gb[1]
        cg = c['enc_coef'] * c['enc_grad_penalty'] * 2.0 / (self.AMBg if self.shard else self.AMB)
        s = math.sqrt(cg)
        nl = len(chain)
        be.enc_gp_seed(e, self.enc_z, self.Ue[:, off:], AMB, self.z, scale=s)
        # the chain: a second data-gradient pass seeded with u
        self._dgrad(head, self.Ue, self.Re[-1], AMB, H[-1], chain[-1].act)
        for l in range(nl - 1, 0, -1):
            self._dgrad(chain[l], self.Re[l], self.Re[l - 1], AMB, H[l - 1], chain[l - 1].act)
        d0 = chain[0]
        # (Ge and everything derived from it - the second operand of the weight-gradient pairs - carries the gradient
        #  scale S, like the back-propagated operand of every other weight gradient)
        S = self.gs
        self._nt(self.Re[0], d0.Wts, self.Ge, AMB, d0.k_pad, d0.n_pad, alpha=S, alpha_dev=self._dS)         # S s * g
        be.sqnorm(self.Ge, AMB, d0.k_pad, self.acc, L.ACC_ENC_GP, scale=1.0 / (cg * S * S), dyn=self._dI2)
        # its backward: forward-shaped launches without bias, masked by the same activations
        x = self.Ge
        for l in range(nl):
            d = chain[l]
            aux, mode = self._aux(H[l], L.AUX_RELU_MASK)
            self._nt(x, d.Ws, self.Qe[l], AMB, d.n_pad, d.k_pad, aux=aux, aux_mode=mode)
            x = self.Qe[l]
        self._nt(x, head.Ws, self.DUe, AMB, head.n_pad, head.k_pad, alpha=s / S, alpha_dev=self._dI)        # du (unscaled)
        # weight gradients (no bias terms: the chain has none)
        for l in range(nl):
            d = chain[l]
            self._tn(self.Re[l], self.Ge if l == 0 else self.Qe[l - 1], d.gW[0], AMB, d.n_pad, d.k_pad, d.N, d.K,
                     d.split_src, d.split_dst)
        for (name, nr, poff), gW in zip(head.parts, head.gW):
            if sep or poff == off:
                self._tn(self.Ue[:, poff:], self.Qe[-1], gW, AMB, P(nr), head.k_pad, nr, head.K, head.split_src, head.split_dst)
        be.enc_gp_back(e, self.enc_z, self.DUe[:, off:], d_e, db, AMB, self.z, grad_scale=S, dyn=self._dS)

    def _gp_scales(self):
        """(Sc, Sr): the gradient scale S (a power of two) split between the two factors of the penalty chain's products, as evenly
        as powers of two allow.  Dynamic scale: Sc stays put (2^6, the static mode's value at S = 4096) and Sr = S / Sc moves with the
        scale - returned here is its HOST factor 1 / Sc, the launches multiply the device's S (_dS), 1 / S (_dI), 1 / S^2 (_dI2) in."""
        split = self.engine_opts.get('gp_scale_split', True)
        if self.dyn_scale:
            sc = 64.0 if split else 1.0
            return sc, 1.0 / sc
        if not split:
            return 1.0, self.gs
        e = int(round(math.log2(self.gs)))
        sc = 2.0 ** ((e + 1) // 2)
        return sc, self.gs / sc

    def _disc_backward(self):
        """Discriminator trunk backward with the gradient penalty riding on the same launches.

        J = coef * mean_rows |d logit / d x_demo|^2 with ReLU layers (learning/amp_agent.py:453-459):
        g_L = m_L * w_logit ; g_{l-1} = m_{l-1} * (g_l @ W_l) ; g_0 = g_1 @ W_1, then the backward of that chain.
        The chain is carried scaled by s = sqrt(2 coef / AMB) so that every weight-gradient term needs alpha = 1:
          rows [3 AMB, 4 AMB) of dZd4[l] hold s*g_l        (data-gradient launches, demo rows' masks via aux row wrap)
          rows [3 AMB, 4 AMB) of Xd4 / Hd4[l] hold s*g_0 / the masked dJ/dU_l   (3 small NT launches)
          gW_l += [dZ_l ; s g_l]^T [H_{l-1} ; dJ/dU_{l-1}]                        (ONE weight-gradient launch per layer)."""
        be, c, AMB = self.be, self.cfg, self.AMB
        Rd = 3 * AMB
        nl = len(self.disc)
        gp_coef = c['disc_coef'] * c['disc_grad_penalty']
        if gp_coef == 0:
            be.zero_(self.G0)     # keeps the reported penalty at 0 without the chain
            self._bwd_chain(self.disc, self.Xd, self.Hd, self.dZd, Rd)
            return
        if self.gp32:
            return self._gp_f32(gp_coef)
        if any(d.act != L.ACT_RELU for d in self.disc):
            return self._disc_backward_curved(gp_coef)
        assert nl >= 2, "gradient penalty with a single discriminator layer is not implemented"
        cg = gp_coef * 2.0 / self.AMBg
        s = math.sqrt(cg)
        top = self.disc[-1]
        # the gradient scale S is split between the two factors of the penalty's weight-gradient products: the chain g_l carries
        # Sc, the dJ/dU side Sr (Sc Sr = S, powers of two).  Both are O(1e-4) quantities - unscaled, the chain sat in half's
        # subnormal range (< 6.1e-5) and the reported penalty came out 0.6-1.5e-4 off; bf16 / f32: S = Sc = Sr = 1.
        Sc, Sr = self._gp_scales()
        be.gp_seed(self.Hd[-1][2 * AMB:], self.disc_head.W[0].view(-1), self.Gp[-1], AMB, top.N, scale=s * Sc)
        # data-gradient chain on 4 AMB rows: [dZ_l ; s g_l] -> [dZ_{l-1} ; s g_{l-1}]
        for l in range(nl - 1, 0, -1):
            d, pl = self.disc[l], self.disc[l - 1]
            aux, mode = self._aux(self.Hd4[l - 1], _AUX[pl.act])
            self._nt(self.dZd4[l], d.Wts, self.dZd4[l - 1], 4 * AMB, d.k_pad, d.n_pad, aux=aux, aux_mode=mode,
                       aux_split=Rd, aux_delta=AMB)
        d0 = self.disc[0]
        # (the chain's second-operand side - G0 and the dJ/dU_l derived from it - carries Sr so that the stacked
        #  weight-gradient launches undo S = Sc Sr for both row blocks with one alpha)
        self._nt(self.Gp[0], d0.Wts, self.G0, AMB, d0.k_pad, d0.n_pad, alpha=Sr / Sc, alpha_dev=self._dS)        # Sr s * g_0
        be.sqnorm(self.G0, AMB, d0.k_pad, self.acc, L.ACC_GP, scale=1.0 / (cg * Sr * Sr), dyn=self._dI2)
        # backward of the chain (values scaled by s; see the docstring): dJ/dU_l, masked by the demo rows' ReLU masks
        aux, mode = self._aux(self.Hd[0][2 * AMB:], L.AUX_RELU_MASK)
        self._nt(self.G0, d0.Ws, self.dGp[0], AMB, d0.n_pad, d0.k_pad, aux=aux, aux_mode=mode)
        for l in range(1, nl):
            d = self.disc[l]
            last = l == nl - 1
            aux, mode = self._aux(self.Hd[l][2 * AMB:], L.AUX_RELU_MASK)
            self._nt(self.dGp[l - 1], d.Ws, self.GpTop if last else self.dGp[l], AMB, d.n_pad, d.k_pad, aux=aux,
                       aux_mode=mode, alpha=s / Sr if last else 1.0, alpha_dev=self._dI if last else None)
            if last:      # the penalty's gradient w.r.t. the logit weights: column sums of the top launch (f32, true scale)
                be.colsum(self.GpTop, AMB, d.N, self.disc_head.gW[0].view(-1))
        # weight (+ bias) gradients: one launch per layer over the stacked rows
        for l in range(nl):
            d = self.disc[l]
            X = self.Xd4 if l == 0 else self.Hd4[l - 1]
            self._tn(self.dZd4[l], X, d.gW[0], 4 * AMB, d.n_pad, d.k_pad, d.N, d.K, d.split_src, d.split_dst,
                     gbias=d.gb[0], bias_rows=Rd)

    def _gp_value(self, amp_streams, gp_coef):
        """VALUE path of the gradient penalty in a gp_f32 engine (see _gp_f32): the demo rows normalised into an f32 input,
        f32 shadows of the trunk, forward (exact ReLU masks), seed, chain g_l, S s g_0 - six f32-storage matrix launches
        ('x3': three f16 MFMAs per product on hi / lo splits of scaled operands).  Reads the branch's master weights and the AMP statistics,
        writes only its own buffers: it runs beside the loss rows' forward (engine_opts gp_stream)."""
        be, AMB, g = self.be, self.AMB, self._gp32
        nl, S = len(self.disc), self.gs
        s = math.sqrt(gp_coef * 2.0 / self.AMBg)
        bits = L.AUX_RELU_BITS
        src, sidx, srm = amp_streams[2]        # the demo stream once more, into the f32 input of the penalty path
        be.rms_normalize(src, self.amp, sidx, srm, AMB, self.amp_mean[2], self.amp_std[2], [g.X])
        # gp_f32 = 'x3': the six f32-storage launches multiply as three 16-bit MFMAs per product on hi / lo splits instead of the
        # exact-f32 MFMA (1/16 of the 16-bit rate): IEEE-half parts of power-of-two scaled operands (ASE_F32H3, unit roundoff
        # ~2^-22; round 4 used bf16 parts, ~2^-17, and the driver's run missed the 1e-4 bar on the penalty by 8 %).  Half's
        # narrow exponent range needs the operands near [2^-2, 2^15] after scaling - and the ranges here are known:
        #   normalised observations  |x| <= 5 (the normaliser's clamp)          2^12
        #   hidden activations       O(1); saturation above 1023                2^6
        #   chain values s g_l       O(1e-3 .. 1e-1); saturation above 16       2^12
        #   weights                  O(1/sqrt(K)); saturation above 32          2^11
        # (engine_opts gp_split = 'bf16' keeps round 4's bf16 split - no range assumption at all - for the same-box A/B)
        x3_prev = getattr(be, 'x3', None)
        mode = self.cfg.get('gp_f32')
        half = mode == 'x3' and x3_prev is not None and self.engine_opts['gp_split'] == 'f16'
        if mode == 'x3' and x3_prev is not None:
            be.x3 = 'f16' if half else True
        ex = (lambda ea: {'x3_exps': (ea, 11)}) if half else (lambda ea: {})
        for l, d in enumerate(self.disc):
            # (half split: the shadows are written PRE-SPLIT - scaled, [8 hi | 8 lo] halves per group of 8 - once per step instead
            #  of being split again by every row tile of the six launches)
            be.refresh_shadow(d.W[0], g.Ws[l], g.Wts[l], d.split_src, d.split_dst, **({'x3_exp': 11} if half else {}))
        x = g.X
        for l, d in enumerate(self.disc):
            self._nt(x, g.Ws[l], g.H[l], AMB, d.n_pad, d.k_pad, bias=d.bs, act=L.ACT_RELU, mask_out=g.bits[l],
                       **ex(12 if l == 0 else 6))
            x = g.H[l]
        top = self.disc[-1]
        be.gp_seed(g.H[-1], self.disc_head.W[0].view(-1), g.Gp[-1], AMB, top.N, scale=s)
        for l in range(nl - 1, 0, -1):
            d = self.disc[l]
            self._nt(g.Gp[l], g.Wts[l], g.Gp[l - 1], AMB, d.k_pad, d.n_pad, aux=g.bits[l - 1], aux_mode=bits, **ex(12))
        d0 = self.disc[0]
        self._nt(g.Gp[0], g.Wts[0], g.G0, AMB, d0.k_pad, d0.n_pad, alpha=S, alpha_dev=self._dS, **ex(12))         # S s * g_0
        if x3_prev is not None:
            be.x3 = x3_prev

    def _gp_f32(self, gp_coef):
        """The gradient penalty of the demo rows (learning/amp_agent.py:453-459) through an exact-f32 value path inside a
        16-bit engine (config gp_f32 / precision 'f16gp32').  The penalty is driven towards zero by training, i.e.
        d logit / d x becomes a CANCELLING sum over the trunk's weights: its relative error in 16-bit storage grows as it
        shrinks (f16: 6e-5 at a penalty of 0.047, 6.6e-4 at 0.0077 - weight rounding first, ReLU mask flips second;
        DESIGN 3.2).  So the VALUE takes nothing from the 16-bit launches: f32 shadows of the trunk, its own forward of the
        AMB demo rows (exact masks), the chain g_l and |g_in|^2 as exact-f32 MFMA launches (_gp_value: 6 launches, ~50 GFLOP
        per step for config 2, submitted with the branch's head).  The chain is then handed to the 16-bit machinery - one
        conversion launch writes s g_l and S s g_0 into the 4th row block of the discriminator's buffers - and the penalty's
        BACKWARD (dJ/dU_l through the exact masks, the logit-weight term, the stacked weight-gradient problems) runs as in
        _disc_backward; the loss rows' data-gradient launches shrink to 3 AMB rows."""
        be, AMB, g = self.be, self.AMB, self._gp32
        Rd, nl, S = 3 * AMB, len(self.disc), self.gs
        cg = gp_coef * 2.0 / self.AMBg
        s = math.sqrt(cg)
        bits = L.AUX_RELU_BITS
        d0 = self.disc[0]
        if self._gp_value_done is not None:
            self._join_branch(self._gp_value_done)
            self._gp_value_done = None
        if getattr(self, '_gp_value_pending', None) is not None:
            self._gp_value(*self._gp_value_pending)
            self._gp_value_pending = None
        be.sqnorm(g.G0, AMB, d0.k_pad, self.acc, L.ACC_GP, scale=1.0 / (cg * S * S), dyn=self._dI2)
        # [dZ_l ; s g_l] and [X ; S s g_0]: the exact chain, rounded once, in the storage type
        if g.cast is None:
            code = {torch.bfloat16: L.BF16, torch.float16: L.F16}[self.dtype]
            items = [(g.Gp[l], self.disc[l].n_pad, self.Gp[l]) for l in range(nl)] + [(g.G0, d0.k_pad, self.G0)]
            rows = [[src.data_ptr(), src.stride(0), c, dst.data_ptr(), dst.stride(0), code] for src, c, dst in items]
            g.cast = (torch.tensor(rows, dtype=torch.int64, device=self.dev), items)
        be.gather_multi(g.cast[0], g.cast[1], None, (0, 0), AMB)
        # the loss rows' data-gradient chain (3 AMB rows)
        for l in range(nl - 1, 0, -1):
            self._dgrad(self.disc[l], self.dZd[l], self.dZd[l - 1], Rd, self.Hd[l - 1], self.disc[l - 1].act)
        # backward of the penalty chain (values carry s and the gradient scale S), through the EXACT masks
        self._nt(self.G0, d0.Ws, self.dGp[0], AMB, d0.n_pad, d0.k_pad, aux=g.bits[0], aux_mode=bits)
        for l in range(1, nl):
            d = self.disc[l]
            last = l == nl - 1
            self._nt(self.dGp[l - 1], d.Ws, self.GpTop if last else self.dGp[l], AMB, d.n_pad, d.k_pad, aux=g.bits[l],
                       aux_mode=bits, alpha=s / S if last else 1.0, alpha_dev=self._dI if last else None)
            if last:
                be.colsum(self.GpTop, AMB, d.N, self.disc_head.gW[0].view(-1))
        for l in range(nl):
            d = self.disc[l]
            X = self.Xd4 if l == 0 else self.Hd4[l - 1]
            self._tn(self.dZd4[l], X, d.gW[0], 4 * AMB, d.n_pad, d.k_pad, d.N, d.K, d.split_src, d.split_dst,
                     gbias=d.gb[0], bias_rows=Rd)

    def _disc_backward_curved(self, gp_coef):
        """Discriminator backward with the gradient penalty for activations with curvature (anything but ReLU; SURVEY 8 row
        X1, learning/amp_agent.py:453-459 with create_graph=True).  Per demo row, with z_l the pre-activations, a'_l = act'(z_l):
            chain      g_L = a'_L * w_logit,  u_{l-1} = W_l^T g_l,  g_{l-1} = a'_{l-1} * u_{l-1},  g_in = W_1^T g_1
            penalty    J = c / AMB  sum |g_in|^2
            its backward: r_0 = W_1 (dJ/dg_in);  dJ/du_l = a'_l * r_l (stored dGp[l]);  r_{l+1} = W_{l+1} (dJ/du_l)
            weights    dJ/dW_l = g_l (dJ/du_{l-1})^T  (the stacked weight-gradient launches, as in the ReLU form)
            NEW        dJ/dz_l += a''_l * u_l * r_l = (a''_l / a'_l^2) * g_l * dGp[l]      (ase_hip_gp_second)
        The last line is the path through a'(z) that vanishes for ReLU; it joins the demo rows of the ordinary backward
        (dZ_l) BEFORE that layer's data- and weight-gradient launches, so the penalty chain (which needs nothing of the loss
        backward) runs first here instead of riding on the discriminator's data-gradient launches.  Scales as in
        _disc_backward: chain values carry s = sqrt(2 c / AMB) and Sc, the r side s and Sr (Sc Sr = the gradient scale S)."""
        be, c, AMB = self.be, self.cfg, self.AMB
        Rd, nl, S = 3 * AMB, len(self.disc), self.gs
        cg = gp_coef * 2.0 / self.AMBg
        s = math.sqrt(cg)
        demo = slice(2 * AMB, 3 * AMB)
        top = self.disc[-1]
        # ---- the chain on the demo rows (AMB-row launches into the 4th row block of dZd4)
        Sc, Sr = self._gp_scales()               # chain side / dJ/dU side of the gradient scale, see _disc_backward
        be.gp_seed(self._twin(self.Hd[-1][demo], top), self.disc_head.W[0].view(-1), self.Gp[-1], AMB, top.N, scale=s * Sc,
                   act=top.act)
        for l in range(nl - 1, 0, -1):
            d, pl = self.disc[l], self.disc[l - 1]
            aux, mode = self._aux(self.Hd[l - 1][demo], _AUX[pl.act])
            self._nt(self.Gp[l], d.Wts, self.Gp[l - 1], AMB, d.k_pad, d.n_pad, aux=aux, aux_mode=mode)
        d0 = self.disc[0]
        self._nt(self.Gp[0], d0.Wts, self.G0, AMB, d0.k_pad, d0.n_pad, alpha=Sr / Sc, alpha_dev=self._dS)        # Sr s * g_in
        be.sqnorm(self.G0, AMB, d0.k_pad, self.acc, L.ACC_GP, scale=1.0 / (cg * Sr * Sr), dyn=self._dI2)
        # ---- its backward: dGp[l] = a'_l * (dGp[l-1] @ W_l^T), the last one only for the logit weights' gradient
        x = self.G0
        for l in range(nl):
            d = self.disc[l]
            last = l == nl - 1
            aux, mode = self._aux(self.Hd[l][demo], _AUX[d.act])
            self._nt(x, d.Ws, self.GpTop if last else self.dGp[l], AMB, d.n_pad, d.k_pad, aux=aux, aux_mode=mode,
                       alpha=s / Sr if last else 1.0, alpha_dev=self._dI if last else None)
            if last:
                be.colsum(self.GpTop, AMB, d.N, self.disc_head.gW[0].view(-1))
            if last:     # the top layer's dGp in storage type and S scale, for the second-order term below
                self._nt(x, d.Ws, self.dGp[l], AMB, d.n_pad, d.k_pad, aux=aux, aux_mode=mode)
            x = self.dGp[l]
        # ---- ordinary backward, the second-order terms joining the demo rows layer by layer
        for l in range(nl - 1, -1, -1):
            d = self.disc[l]
            be.gp_second(self._twin(self.Hd[l][demo], d), self.Gp[l], self.dGp[l], self.dZd[l][demo], AMB, d.N, d.act)
            X = self.Xd4 if l == 0 else self.Hd4[l - 1]
            self._tn(self.dZd4[l], X, d.gW[0], 4 * AMB, d.n_pad, d.k_pad, d.N, d.K, d.split_src, d.split_dst,
                     gbias=d.gb[0], bias_rows=Rd)
            if l > 0:
                pl = self.disc[l - 1]
                self._dgrad(d, self.dZd[l], self.dZd[l - 1], Rd, self.Hd[l - 1], pl.act)

    # ------------------------------------------------------------------ collectives (single rank: no-ops)
    def _host(self, fn):
        """A torch-level operation INSIDE the step (the collectives of the exchange points and the few tensor operations
        around them): executed now on the current stream, or - while the backend records a launch program - recorded as a
        host callback at this position of the sequence and executed on every replay, on the stream that is current NOW
        (kernel launches are not the only things a replay must repeat: a program that dropped them would exchange nothing)."""
        if self.dev.type != 'cuda':
            self.be.host_call(fn)
            return
        s = torch.cuda.current_stream(self.dev)

        def run():
            with torch.cuda.stream(s):
                fn()
        self.be.host_call(run)

    def _ar(self, t):
        """SUM all-reduce of a device tensor over the data-parallel group: RCCL over xGMI in production (backend
        'nccl').  With the 'gloo' backend (CPU test rigs, or several ranks sharing one GPU) device tensors are staged
        through the host."""
        import torch.distributed as dist

        def run():
            if t.is_cuda and dist.get_backend() == 'gloo':
                h = t.cpu()
                dist.all_reduce(h)
                t.copy_(h)
            else:
                dist.all_reduce(t)
        self._host(run)

    def _exchange_bucket(self, group, lo, hi, with_acc=False):
        """SUM exchange of one gradient bucket (+ with_acc: the step's loss partial sums acc[1 : ACC_LOGIT_W2], f64, travelling as
        (hi, lo) f32 pairs: every rank's value is carried to ~2^-48, the all-reduce ADDS in f32, so the exchanged sums are good to
        ~2^-24 relative - they are reported scalars and the kl of the adaptive schedule, no gradient depends on them) as ONE collective.  f32 payload without the sums: in place, no copy.
        Otherwise through a persistent exchange buffer inside the host callback: pack (+ convert: dp_grad_dtype 'bf16' halves the
        bytes on the links) -> all-reduce -> unpack; two device copies of the bucket against an xGMI ring pass of it."""
        g = self.grads[lo:hi]
        if self.dp_grad_dtype == 'f32' and not with_acc:
            self._ar(g)
            return
        n, k = g.numel(), (L.ACC_LOGIT_W2 - 1) if with_acc else 0
        dt = torch.float32 if self.dp_grad_dtype == 'f32' else torch.bfloat16
        key = (group, dt, k)
        if key not in self._xbuf:
            self._xbuf[key] = torch.zeros(n + 2 * k, dtype=dt, device=self.dev)
        xb, acc = self._xbuf[key], self.acc
        import torch.distributed as dist

        def run():
            xb[:n].copy_(g)
            if k:
                a = acc[1:1 + k]
                hi_ = a.float()
                xb[n:n + k].copy_(hi_)
                xb[n + k:].copy_((a - hi_.double()).float())
            if xb.is_cuda and dist.get_backend() == 'gloo':
                h = xb.cpu()
                dist.all_reduce(h)
                xb.copy_(h)
            else:
                dist.all_reduce(xb)
            g.copy_(xb[:n])
            if k:
                acc[1:1 + k] = xb[n:n + k].double() + xb[n + k:].double()
        self._host(run)

    def _dist_on(self):
        return self.R > 1 or self.force_dist

    def _allreduce_stats(self):
        """Exchange of the step's partial statistics when it is NOT done inside the un-chained heads (_stats_exchanged): the
        observation sums and the mask sum here; the amp block too unless the discriminator's head - which computes it on its own
        stream whenever streams exist - has exchanged it (_exchange_amp_sums)."""
        if self._dist_shard() and not self._stats_exchanged:
            if self._prep is not None:
                self._join_branch(self._prep)          # (short prologue: the mask sum was formed on the critic's stream)
            if self.masked:
                self.be.copy_(self.stats_flat[-1:], self.acc[L.ACC_MASK_SUM:L.ACC_MASK_SUM + 1])
            n_amp = self.amp_sums_flat.numel() if (self.has_disc and self._amp_stats_in_branch()) else 0
            self._ar(self.stats_flat[n_amp:])
            if self.masked:
                self.be.copy_(self.acc[L.ACC_MASK_SUM:L.ACC_MASK_SUM + 1], self.stats_flat[-1:])

    def _allreduce_grads(self):
        if self._dist_on():
            self._exchange_bucket('all', 0, self.n_train)     # SUM of per-rank partials (global denominators inside)
            if self.shard:
                self._ar(self.acc[1:])              # slot 0 (mask sum) is already global
            else:
                # Horovod semantics: every rank's loss is complete on its own minibatch; the optimizer sees the AVERAGE
                self._host(lambda: self.grads[:self.n_train].mul_(1.0 / self.R))

    def _average_kl(self):
        """Horovod mode: the step's kl is averaged over the ranks before it is reported and before the adaptive schedule sees
        it (learning/amp_agent.py:224-228, learning/common_agent.py:204-208: hvd.average_value(curr_train_info['kl']) under the
        default 'legacy' schedule type) - every rank then derives the SAME learning rate from it; with local kls the ranks
        would apply the averaged gradient with different rates and their parameters drift apart.  (Sharded mode: the kl sum
        is part of the accumulator all-reduce already.)"""
        if self._dist_on() and not self.shard:
            kl = self.acc[L.ACC_KL:L.ACC_KL + 1]
            self._ar(kl)
            self._host(lambda: kl.mul_(1.0 / self.R))

    def _identity_stats(self, mean, std):
        self._host(lambda: (mean.zero_(), std.fill_(1.0)))

    # ------------------------------------------------------------------ results
    def results(self, snapshot=False):
        """train_result with the reference's keys (learning/ase_agent.py:296-306) as device scalars.  snapshot=True:
        views of ONE copy of the scalar vector (+ one of the logit column), instead of live views the next step
        overwrites: two small device copies per optimisation step."""
        r = self.res.clone() if snapshot else self.res
        r = r.view(-1)
        out = {'entropy': r[L.RES_ENTROPY], 'kl': r[L.RES_KL], 'b_loss': r[L.RES_B_LOSS], 'actor_loss': r[L.RES_A_LOSS],
               'actor_clip_frac': r[L.RES_CLIP_FRAC], 'critic_loss': r[L.RES_C_LOSS], 'loss': r[L.RES_LOSS]}
        if self.has_disc:
            AMB = self.AMB
            logit = self.logit_slot.clone() if snapshot else self.logit_slot       # (written by the branch's head, _disc_forward)
            out.update({'disc_loss': r[L.RES_DISC_LOSS], 'disc_grad_penalty': r[L.RES_DISC_GP],
                        'disc_logit_loss': r[L.RES_DISC_LOGIT_LOSS], 'disc_agent_acc': r[L.RES_DISC_AGENT_ACC],
                        'disc_demo_acc': r[L.RES_DISC_DEMO_ACC], 'disc_agent_logit': logit[:2 * AMB],
                        'disc_demo_logit': logit[2 * AMB:]})
        if self.has_enc:
            out['enc_loss'] = r[L.RES_ENC_LOSS]
            if self.enc_gp:
                out['enc_grad_penalty'] = r[L.RES_ENC_GP]
        if self.div_on:
            out['amp_diversity_loss'] = r[L.RES_DIV_LOSS]
        return out

    # ------------------------------------------------------------------ inference (rollout side, epoch tail)
    def _scr(self, key, rows, cols, dt=None):
        dt = self.dtype if dt is None else dt
        k = (key, cols, dt)
        t = self._scratch.get(k)
        if t is None or t.shape[0] < rows:
            t = torch.zeros(rows, cols, dtype=dt, device=self.dev)
            self._scratch[k] = t
        return t[:rows]

    def _eval_stats(self, state, D, key):
        mean, std = self._scr(key + '_m', 1, D, torch.float32), self._scr(key + '_s', 1, D, torch.float32)
        self.be.rms_finalize(state, D, None, 0, 0, mean, std)
        return mean[0], std[0]

    def _policy_nets(self, obs, z, normalize, want_mu, want_value):
        """Eval-mode observation normalisation + actor / critic forward on raw observations [n, obs].  Returns the padded
        head outputs (MU f32 [n, P(act)] before any tanh, V f32 [n, 64]); every buffer is persistent scratch."""
        be, n = self.be, obs.shape[0]
        f32 = torch.float32
        a0 = self.actor[0]
        Xa, Xc = self._scr('Xa', n, a0.k_pad), self._scr('Xc', n, a0.k_pad)
        if normalize and self.cfg.get('normalize_input', True):
            mean, std = self._eval_stats(self.obs_state, self.obs, 'obs')
        else:
            mean, std = self._scr('one_m', 1, self.obs, f32)[0].zero_(), self._scr('one_s', 1, self.obs, f32)[0].fill_(1.0)
        be.rms_normalize(obs, self.obs, None, (0, 0), n, mean, std, [Xa, Xc])
        MU = V = None
        if self.z:
            sd = a0.split_dst
            be.gather_rows(z, self.z, None, (0, 0), n, Xc[:, sd:])
        if want_mu:
            if self.style:
                Zs = self._scr('Zs', n, P(self.z))
                be.gather_rows(z, self.z, None, (0, 0), n, Zs)
                h = Zs
                for i, d in enumerate(self.style[:-1]):
                    y = self._scr(f'Hs{i}', n, d.n_pad)
                    self._fwd(d, h, y, n)
                    h = y
                self._fwd(self.style[-1], h, Xa[:, a0.split_dst:], n)
            h = Xa
            for i, d in enumerate(self.actor):
                y = self._scr(f'Ha{i}', n, d.n_pad)
                self._fwd(d, h, y, n)
                h = y
            MU = self._scr('MU', n, self.mu_head.n_pad, f32)
            self._fwd(self.mu_head, h, MU, n)
        if want_value:
            h = Xc
            for i, d in enumerate(self.critic):
                y = self._scr(f'Hc{i}', n, d.n_pad)
                self._fwd(d, h, y, n)
                h = y
            V = self._scr('V', n, self.value_head.n_pad, f32)
            self._fwd(self.value_head, h, V, n)
        return MU, V

    def _value_out(self, V, n, unnorm_value, key='value_out'):
        """Column 0 of the padded value head -> dense [n, 1] (un-normalised like RunningMeanStd(unnorm=True))."""
        v = self._scr(key, n, 1, torch.float32)
        self.be.gather_rows(V, 1, None, (0, 0), n, v)
        if unnorm_value and self.norm_value:
            self.be.rms_unnormalize(self.val_state, v, v)
        return v

    def policy_forward(self, obs, z=None, normalize=True, unnorm_value=True, want=('mu', 'value')):
        """Eval-mode actor / critic on raw observations [n, obs] (learning/ase_agent.py:117-148,385-393):
        eval-mode obs normalisation -> nets -> (mu [n, act], value [n, 1] un-normalised).  The returned tensors are
        persistent scratch, overwritten by the next call with the same n."""
        n = obs.shape[0]
        MU, V = self._policy_nets(obs, z, normalize, 'mu' in want, 'value' in want)
        out = {}
        if 'mu' in want:
            mu = self._scr('mu_out', n, self.act, torch.float32)
            self.be.gather_rows(MU, self.act, None, (0, 0), n, mu)
            out['mu'] = torch.tanh_(mu) if self.mu_tanh else mu
        if 'value' in want:
            out['value'] = self._value_out(V, n, unnorm_value)
        return out

    def policy_act(self, obs, z, rand_probs, rng_state):
        """One rollout step of get_action_values (learning/amp_agent.py:139-169, learning/ase_agent.py:117-148) as an
        explicit launch sequence: normalise -> actor + critic -> ase_hip_sample_actions (tanh / Normal sample / neglogp /
        eps-greedy on device) -> value un-normalisation.  rand_probs None: every row stochastic (plain PPO)."""
        n, f32 = obs.shape[0], torch.float32
        MU, V = self._policy_nets(obs, z, True, True, True)
        o = {k: self._scr('act_' + k, n, c, f32) for k, c in (('mus', self.act), ('sigmas', self.act), ('actions', self.act),
                                                             ('neglogpacs', 1), ('rand_action_mask', 1))}
        self.be.sample_actions(MU, self.logstd, rand_probs, rng_state, o['mus'], o['sigmas'], o['actions'],
                               o['neglogpacs'], o['rand_action_mask'] if rand_probs is not None else None, n, self.act,
                               self.mu_tanh)
        res = {'neglogpacs': o['neglogpacs'].view(-1), 'values': self._value_out(V, n, True, 'act_values'),
               'actions': o['actions'], 'mus': o['mus'], 'sigmas': o['sigmas'], 'rnn_states': None}
        if rand_probs is not None:
            res['rand_action_mask'] = o['rand_action_mask'].view(-1)
        return res

    def amp_heads(self, amp_obs, normalize=True):
        """Eval-mode discriminator logits [n,1] (+ un-normalised encoder output [n,z]) on raw amp obs
        (learning/amp_agent.py:547-549, learning/ase_agent.py:480-482)."""
        be, n = self.be, amp_obs.shape[0]
        f32 = torch.float32
        d0 = self.disc[0]
        X = self._scr('Xd', n, d0.k_pad)
        if normalize and self.cfg.get('normalize_amp_input', True):
            mean, std = self._eval_stats(self.amp_state, self.amp, 'amp')
        else:
            mean, std = self._scr('one_am', 1, self.amp, f32)[0].zero_(), self._scr('one_as', 1, self.amp, f32)[0].fill_(1.0)
        be.rms_normalize(amp_obs, self.amp, None, (0, 0), n, mean, std, [X])
        h = X
        for i, d in enumerate(self.disc):
            y = self._scr(f'Hd{i}', n, d.n_pad)
            self._fwd(d, h, y, n)
            h = y
        HD = self._scr('HD', n, self.disc_head.n_pad, f32)
        self._fwd(self.disc_head, h, HD, n)
        enc = None
        if self.has_enc:
            if self.enc_sep:
                h = X
                for i, d in enumerate(self.enc_chain):
                    y = self._scr(f'He{i}', n, d.n_pad)
                    self._fwd(d, h, y, n)
                    h = y
                E = self._scr('E', n, self.enc_head.n_pad, f32)
                self._fwd(self.enc_head, h, E, n)
                enc = E
            else:
                enc = HD[:, self.disc_head.parts[1][2]:]
        return HD, enc

    # ------------------------------------------------------------------ once-per-epoch rollout tail
    def prepare_epoch(self, exp):
        """Tail of play_steps + prepare_dataset on the time-major experience buffers (device tensors):
        AMP/ASE rewards, GAE, advantage normalisation, value normalisation (two statistics updates:
        values, then returns — learning/common_agent.py:323-325).  Returns the dataset dict in physical
        (time-major) row order; minibatch rows are addressed with remap = (H, N)."""
        be, c = self.be, self.cfg
        H, N = exp['rewards'].shape[0], exp['rewards'].shape[1]
        B = H * N
        f32 = torch.float32
        r_disc = r_enc = None
        info = {}
        if self.has_disc:
            amp = exp['amp_obs'].view(B, self.amp)
            # Sharded data parallel: the experience buffer is replicated, so the 0.8 TFLOP of discriminator / encoder
            # inference is split by rows and the two reward vectors are assembled by ONE sum all-reduce of a buffer that
            # is zero outside the rank's rows (exact: x + 0).  Everything after it (GAE, moments) is cheap and replicated.
            split = self.shard and self._dist_on() and B % self.R == 0
            lo, n = (self.rank * (B // self.R), B // self.R) if split else (0, B)
            rw = self._scr('r_amp', 2 * B, 1, f32)
            if split:
                be.zero_(rw)
            r_disc = rw[:B]
            HD, enc = self.amp_heads(amp[lo:lo + n])
            be.disc_reward(HD, r_disc[lo:lo + n], n, c['disc_reward_scale'])
            info['disc_rewards'] = r_disc
            if self.has_enc:
                r_enc = rw[B:]
                be.enc_reward(enc, exp['ase_latents'].view(B, self.z)[lo:lo + n], r_enc[lo:lo + n], n, self.z,
                              c['enc_reward_scale'])
                info['enc_rewards'] = r_enc
            if split:
                self._ar(rw)
        w_task = c.get('task_reward_w', 1.0) if self.has_disc else 1.0
        if not self.has_disc and 'disc_rewards' in exp:
            # HRL high-level update: the rollout recorded the frozen low-level controller's discriminator reward
            # (learning/hrl_agent.py:145-147,251-256): r = task_reward_w * r_task + disc_reward_w * r_disc
            r_disc = exp['disc_rewards'].view(B, 1)
            info['disc_rewards'] = r_disc
            w_task = c.get('task_reward_w', 1.0)
        advs, rets = self._scr('advs', B, 1, f32), self._scr('rets', B, 1, f32)
        be.gae(exp['dones'], exp['values'], exp['next_values'], exp['rewards'], r_disc, r_enc,
               w_task, c.get('disc_reward_w', 0.0) if r_disc is not None else 0.0,
               c.get('enc_reward_w', 0.0), c['gamma'], c['tau'], advs, rets, H, N)
        info['mb_advs'], info['mb_returns'] = advs, rets
        values = exp['values'].view(B, 1)
        mask = exp['rand_action_mask'].view(B, 1) if self.masked else None
        adv = self._scr('adv_n', B, 1, f32)
        acc3 = self._scr('acc3', 1, 3, torch.float64)[0]
        be.zero_(acc3)
        be.adv_norm(rets, values, mask, adv, acc3, B, c['normalize_advantage'], 0)
        be.adv_norm(rets, values, mask, adv, acc3, B, c['normalize_advantage'], 1)
        if self.norm_value:
            nv, nr = self._scr('val_n', B, 1, f32), self._scr('ret_n', B, 1, f32)
            sums = self._scr('val_sums', 1, 2, torch.float64)[0]
            m, s = self._scr('val_m', 1, 1, f32), self._scr('val_s', 1, 1, f32)
            for src, dst in ((values, nv), (rets, nr)):
                be.zero_(sums)
                be.rms_moments(src, 1, None, (0, 0), B, self.val_state, sums)
                # (sharded mode: the experience buffer is replicated, every rank computes the same moments; Horovod
                #  mode: local moments, the running statistics are averaged once per epoch by CommonAgent.sync_stats)
                be.rms_finalize(self.val_state, 1, sums, B, 1, m, s)
                be.rms_normalize(src, 1, None, (0, 0), B, m[0], s[0], [dst])
        else:
            nv, nr = values, rets
        ds = {'obs': exp['obses'].view(B, self.obs), 'actions': exp['actions'].view(B, self.act),
              'mu': exp['mus'].view(B, self.act), 'sigma': exp['sigmas'].view(B, self.act),
              'old_logp_actions': exp['neglogpacs'].view(B, 1), 'advantages': adv, 'old_values': nv, 'returns': nr}
        if self.masked:
            ds['rand_action_mask'] = exp['rand_action_mask'].view(B, 1)
            ds['amp_obs'] = exp['amp_obs'].view(B, self.amp)
        if self.z:
            ds['ase_latents'] = exp['ase_latents'].view(B, self.z)
        return ds, info, (H, N)

    def export_grads(self):
        return {name: self.grads[o:o + math.prod(shp)].view(shp) for name, (o, shp) in self.net.param_slices.items()
                if o < self.n_train}
