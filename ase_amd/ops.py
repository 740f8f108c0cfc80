"""``torch.ops.ase_hip.*``: PyTorch custom operators over the C ABI of libase_hip.so (include/ase_hip.h).

This is the binding a maintainer of the reference adds to call the HIP kernels from ordinary PyTorch code: the
operators take / return ``torch.Tensor`` (device-resident), allocate their outputs with torch's allocator, launch on torch's
current HIP stream, never synchronise, and raise ``RuntimeError`` on bad shapes / dtypes / devices.  They are registered for
the CUDA (= HIP) dispatch key only: on CPU tensors PyTorch itself raises ``NotImplementedError`` - there is no fallback.

  linear_act(x, w, b, act) -> y                      fused Linear + activation (learning/ase_network_builder.py:255-259), MFMA GEMM;
                                                      differentiable: backward = linear_bwd_data + linear_bwd_weight, so an
                                                      autograd-based agent (the reference's own ``calc_gradients``) trains through it
  linear_bwd_data(dz, w) -> dx                       dz @ w
  linear_bwd_weight(dz, x) -> (gw, gb)               dz^T @ x, column sums of dz
  rms_update_normalize(x, state) -> y                RunningMeanStd.forward in train mode (state f64 [2 D + 1] updated in place)
  rms_normalize(x, state) -> y                       ... in eval mode
  rms_unnormalize(x, state) -> y                     RunningMeanStd(..., unnorm=True) on values
  gae(dones, values, next_values, rewards, gamma, tau) -> (advs, returns)         CommonAgent.discount_values (+ returns)
  masked_norm(returns, values, mask) -> advantages   AMPAgent._calc_advs (torch_ext.normalization_with_masks)
  gather_rows(src, idx) -> out                       AMPDataset._get_item
  disc_reward(logits, scale) / enc_reward(enc, z, scale)                          AMPAgent._calc_disc_rewards / ASEAgent._calc_enc_rewards
  normalize_rows(x) -> y                             torch.nn.functional.normalize(x, dim=-1)
  sample_latents(n, dim, rng_state) -> z             ASEBuilder.Network.sample_latents (Philox stream {seed, offset}, advanced)
  fused_adam_(w, g, m, v, opt_state)                 torch.optim.Adam.step on flat buffers (in place)
  ppo_loss_head(mu, value, actions, old_mu, old_sigma, old_neglogp, advantages, old_values, returns, mask, logstd, ...)
      -> (stats [6], d_mu, d_value)                   CommonAgent._actor_loss / _critic_loss / bound_loss + entropy, clip fraction, kl
                                                      (learning/common_agent.py:456-464,505-534) with the gradient of
                                                      actor_loss + bounds_coef bound_loss w.r.t. mu and of critic_coef critic_loss w.r.t. value
  disc_loss_gp(logits, grad_demo, disc_coef) -> (stats [4], d_logits)
                                                      AMPAgent._disc_loss' data terms (learning/amp_agent.py:442-459,481-496): BCE halves,
                                                      accuracies, the gradient penalty mean |d logit / d obs|^2, d loss / d logits
  enc_div_loss(enc_pred, enc_z, mu, mu_div, z, z_new, enc_coef, div_coef, div_tar)
      -> (stats [2], d_enc, d_mu, d_mu_div)            ASEAgent._enc_loss + _diversity_loss (learning/ase_agent.py:413-418,445-467)

``HipLinear`` is an ``nn.Linear`` whose forward is ``linear_act`` (optionally with a fused ReLU / tanh).
"""
import torch
import torch.nn as nn

from . import lib as L

_ACT = {'none': L.ACT_NONE, 'None': L.ACT_NONE, 'relu': L.ACT_RELU, 'tanh': L.ACT_TANH}
_be = None


def _backend():
    global _be
    if _be is None:
        from .backend import HipBackend
        _be = HipBackend(torch.device('cuda', torch.cuda.current_device()))
    return _be


def _check(cond, msg):
    if not cond:
        raise RuntimeError('ase_hip: ' + msg)


def _P(x, m):
    return (int(x) + m - 1) // m * m


def _padded(x, cols, dtype=None):
    """Contiguous copy of x [R, C] with the row length padded to `cols` (zeros)."""
    dtype = x.dtype if dtype is None else dtype
    if x.shape[1] == cols and x.is_contiguous() and x.dtype == dtype:
        return x
    out = torch.zeros(x.shape[0], cols, dtype=dtype, device=x.device)
    out[:, :x.shape[1]] = x
    return out


def _gemm_dtype(x):
    _check(x.dtype in (torch.bfloat16, torch.float16, torch.float32), f'matrix operands must be bf16, f16 or f32, got {x.dtype}')
    return x.dtype


def _kpad(k, dtype):
    return _P(k, 32 if dtype in (torch.bfloat16, torch.float16) else 16)          # whole 64-byte K steps


# ------------------------------------------------------------------------------------------------ dense layers
@torch.library.custom_op('ase_hip::linear_act', mutates_args=(), device_types='cuda')
def linear_act(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, act: str) -> torch.Tensor:
    _check(x.dim() == 2 and w.dim() == 2 and x.shape[1] == w.shape[1], 'linear_act: x [M, K], w [N, K]')
    _check(act in _ACT, f'linear_act: activation {act!r} (none | relu | tanh)')
    dt = _gemm_dtype(x)
    M, K = x.shape
    N = w.shape[0]
    kp, npad = _kpad(K, dt), _P(N, 64)
    xs, ws = _padded(x, kp), _padded(w.to(dt), kp)
    if npad != N:
        ws = torch.cat([ws, torch.zeros(npad - N, kp, dtype=dt, device=x.device)])
    bs = torch.zeros(npad, dtype=torch.float32, device=x.device)
    bs[:N] = b.float()
    y = torch.empty(M, npad, dtype=dt, device=x.device)
    _backend().gemm_nt(xs, ws, y, M, npad, kp, bias=bs, act=_ACT[act])
    return y[:, :N].contiguous() if npad != N else y


@linear_act.register_fake
def _(x, w, b, act):
    return x.new_empty(x.shape[0], w.shape[0])


@torch.library.custom_op('ase_hip::linear_bwd_data', mutates_args=(), device_types='cuda')
def linear_bwd_data(dz: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    _check(dz.dim() == 2 and w.dim() == 2 and dz.shape[1] == w.shape[0], 'linear_bwd_data: dz [M, N], w [N, K]')
    dt = _gemm_dtype(dz)
    M, N = dz.shape
    K = w.shape[1]
    npad, kout = _kpad(N, dt), _P(K, 64)
    wt = _padded(w.to(dt).t(), npad)                               # [K, N]: the contraction dim contiguous in both operands
    if kout != K:
        wt = torch.cat([wt, torch.zeros(kout - K, npad, dtype=dt, device=dz.device)])
    dx = torch.empty(M, kout, dtype=dt, device=dz.device)
    _backend().gemm_nt(_padded(dz, npad), wt, dx, M, kout, npad)
    return dx[:, :K].contiguous() if kout != K else dx


@linear_bwd_data.register_fake
def _(dz, w):
    return dz.new_empty(dz.shape[0], w.shape[1])


@torch.library.custom_op('ase_hip::linear_bwd_weight', mutates_args=(), device_types='cuda')
def linear_bwd_weight(dz: torch.Tensor, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    _check(dz.dim() == 2 and x.dim() == 2 and dz.shape[0] == x.shape[0], 'linear_bwd_weight: dz [M, N], x [M, K]')
    dt = _gemm_dtype(dz)
    _check(x.dtype == dt, 'linear_bwd_weight: dz and x must have the same dtype')
    M, N = dz.shape
    K = x.shape[1]
    el = 8 if dt in (torch.bfloat16, torch.float16) else 4
    npad, kp = _P(N, el), _P(K, el)
    gw = torch.zeros(N, K, dtype=torch.float32, device=dz.device)
    gb = torch.zeros(N, dtype=torch.float32, device=dz.device)
    _backend().gemm_tn(_padded(dz, npad), _padded(x, kp), gw, M, npad, kp, N, K, K, K, gbias=gb)
    return gw, gb


@linear_bwd_weight.register_fake
def _(dz, x):
    return dz.new_empty(dz.shape[1], x.shape[1], dtype=torch.float32), dz.new_empty(dz.shape[1], dtype=torch.float32)


def _linear_setup(ctx, inputs, output):
    x, w, b, act = inputs
    ctx.act = act
    ctx.save_for_backward(x, w, output)


def _linear_backward(ctx, dy):
    x, w, y = ctx.saved_tensors
    dy = dy.contiguous()
    if ctx.act == 'relu':
        dz = dy * (y > 0).to(dy.dtype)
    elif ctx.act == 'tanh':
        dz = dy * (1 - y.float() * y.float()).to(dy.dtype)
    else:
        dz = dy
    dz = dz.to(x.dtype)
    dx = linear_bwd_data(dz, w) if ctx.needs_input_grad[0] else None
    gw, gb = linear_bwd_weight(dz, x)
    return dx, gw.to(w.dtype), gb, None


linear_act.register_autograd(_linear_backward, setup_context=_linear_setup)


class HipLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict) whose forward is ase_hip::linear_act with a fused activation."""

    def __init__(self, in_features, out_features, activation='none', compute_dtype=torch.bfloat16, **kw):
        super().__init__(in_features, out_features, **kw)
        self.activation, self.compute_dtype = activation, compute_dtype

    def forward(self, x):
        shp = x.shape
        y = linear_act(x.reshape(-1, shp[-1]).to(self.compute_dtype), self.weight, self.bias, self.activation)
        return y.reshape(*shp[:-1], -1).to(x.dtype)


# ------------------------------------------------------------------------------------------------ normaliser
def _rms(x, state, update):
    _check(x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous(), 'rms: x must be a contiguous f32 [M, D] tensor')
    M, D = x.shape
    _check(state.dtype == torch.float64 and state.numel() == 2 * D + 1, 'rms: state must be f64 [2 D + 1] = mean | var | count')
    be = _backend()
    mean = torch.empty(1, D, dtype=torch.float32, device=x.device)
    std = torch.empty(1, D, dtype=torch.float32, device=x.device)
    if update:
        sums = torch.zeros(2 * D, dtype=torch.float64, device=x.device)
        be.rms_moments(x, D, None, (0, 0), M, state, sums)
        be.rms_finalize(state, D, sums, M, 1, mean, std)
    else:
        be.rms_finalize(state, D, None, 0, 0, mean, std)
    y = torch.empty_like(x)
    be.rms_normalize(x, D, None, (0, 0), M, mean[0], std[0], [y])
    return y


@torch.library.custom_op('ase_hip::rms_update_normalize', mutates_args=('state',), device_types='cuda')
def rms_update_normalize(x: torch.Tensor, state: torch.Tensor) -> torch.Tensor:
    return _rms(x, state, True)


@rms_update_normalize.register_fake
def _(x, state):
    return torch.empty_like(x)


@torch.library.custom_op('ase_hip::rms_normalize', mutates_args=(), device_types='cuda')
def rms_normalize(x: torch.Tensor, state: torch.Tensor) -> torch.Tensor:
    return _rms(x, state, False)


@rms_normalize.register_fake
def _(x, state):
    return torch.empty_like(x)


@torch.library.custom_op('ase_hip::rms_unnormalize', mutates_args=(), device_types='cuda')
def rms_unnormalize(x: torch.Tensor, state: torch.Tensor) -> torch.Tensor:
    _check(x.dtype == torch.float32 and x.is_contiguous() and state.dtype == torch.float64 and state.numel() == 3,
           'rms_unnormalize: x f32 contiguous, state f64 [3]')
    y = torch.empty_like(x)
    _backend().rms_unnormalize(state, x, y)
    return y


@rms_unnormalize.register_fake
def _(x, state):
    return torch.empty_like(x)


# ------------------------------------------------------------------------------------------------ rollout tail
@torch.library.custom_op('ase_hip::gae', mutates_args=(), device_types='cuda')
def gae(dones: torch.Tensor, values: torch.Tensor, next_values: torch.Tensor, rewards: torch.Tensor, gamma: float,
        tau: float) -> tuple[torch.Tensor, torch.Tensor]:
    _check(rewards.dim() == 3 and rewards.shape[2] == 1 and values.shape == rewards.shape and next_values.shape == rewards.shape,
           'gae: values / next_values / rewards must be [H, N, 1]')
    H, N = rewards.shape[0], rewards.shape[1]
    advs, rets = torch.empty_like(rewards), torch.empty_like(rewards)
    _backend().gae(dones.to(torch.uint8).contiguous(), values.contiguous(), next_values.contiguous(), rewards.contiguous(),
                   None, None, 1.0, 0.0, 0.0, gamma, tau, advs, rets, H, N)
    return advs, rets


@gae.register_fake
def _(dones, values, next_values, rewards, gamma, tau):
    return torch.empty_like(rewards), torch.empty_like(rewards)


@torch.library.custom_op('ase_hip::masked_norm', mutates_args=(), device_types='cuda')
def masked_norm(returns: torch.Tensor, values: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    n = returns.numel()
    _check(values.numel() == n and mask.numel() == n, 'masked_norm: returns / values / mask must have the same number of rows')
    be = _backend()
    adv = torch.empty(n, 1, dtype=torch.float32, device=returns.device)
    acc3 = torch.zeros(3, dtype=torch.float64, device=returns.device)
    r, v, m = returns.reshape(n, 1).contiguous(), values.reshape(n, 1).contiguous(), mask.reshape(n, 1).float().contiguous()
    be.adv_norm(r, v, m, adv, acc3, n, True, 0)
    be.adv_norm(r, v, m, adv, acc3, n, True, 1)
    return adv.view(-1)


@masked_norm.register_fake
def _(returns, values, mask):
    return returns.new_empty(returns.numel())


@torch.library.custom_op('ase_hip::gather_rows', mutates_args=(), device_types='cuda')
def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    _check(src.dim() == 2 and src.dtype == torch.float32 and idx.dtype == torch.int32, 'gather_rows: src f32 [R, D], idx int32 [M]')
    out = torch.empty(idx.numel(), src.shape[1], dtype=torch.float32, device=src.device)
    _backend().gather_rows(src, src.shape[1], idx.contiguous(), (0, 0), idx.numel(), out)
    return out


@gather_rows.register_fake
def _(src, idx):
    return src.new_empty(idx.numel(), src.shape[1])


@torch.library.custom_op('ase_hip::disc_reward', mutates_args=(), device_types='cuda')
def disc_reward(logits: torch.Tensor, scale: float) -> torch.Tensor:
    lg = logits.reshape(-1, 1).float().contiguous()
    r = torch.empty_like(lg)
    _backend().disc_reward(lg, r, lg.shape[0], scale)
    return r.view(logits.shape)


@disc_reward.register_fake
def _(logits, scale):
    return torch.empty_like(logits, dtype=torch.float32)


@torch.library.custom_op('ase_hip::enc_reward', mutates_args=(), device_types='cuda')
def enc_reward(enc: torch.Tensor, z: torch.Tensor, scale: float) -> torch.Tensor:
    _check(enc.shape == z.shape and enc.dim() == 2, 'enc_reward: enc and z must be [n, z_dim]')
    n, D = enc.shape
    r = torch.empty(n, 1, dtype=torch.float32, device=enc.device)
    _backend().enc_reward(enc.float().contiguous(), z.float().contiguous(), r, n, D, scale)
    return r


@enc_reward.register_fake
def _(enc, z, scale):
    return enc.new_empty(enc.shape[0], 1, dtype=torch.float32)


@torch.library.custom_op('ase_hip::normalize_rows', mutates_args=(), device_types='cuda')
def normalize_rows(x: torch.Tensor) -> torch.Tensor:
    _check(x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] <= 128, 'normalize_rows: x f32 [n, d <= 128]')
    xc = x.contiguous()
    y = torch.empty_like(xc)
    _backend().normalize_rows(xc, y, xc.shape[0], xc.shape[1])
    return y


@normalize_rows.register_fake
def _(x):
    return torch.empty_like(x)


@torch.library.custom_op('ase_hip::sample_latents', mutates_args=('rng_state',), device_types='cuda')
def sample_latents(n: int, dim: int, rng_state: torch.Tensor) -> torch.Tensor:
    _check(rng_state.dtype == torch.int64 and rng_state.numel() == 2, 'sample_latents: rng_state int64 [2] = seed | offset')
    z = torch.empty(n, dim, dtype=torch.float32, device=rng_state.device)
    _backend().sample_latents(z, n, dim, rng_state)
    return z


@sample_latents.register_fake
def _(n, dim, rng_state):
    return rng_state.new_empty(n, dim, dtype=torch.float32)


@torch.library.custom_op('ase_hip::fused_adam_', mutates_args=('w', 'm', 'v', 'opt_state'), device_types='cuda')
def fused_adam_(w: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, opt_state: torch.Tensor) -> None:
    """opt_state: f64 [8] = step, lr, beta1, beta2, eps, bias_corr1, bias_corr2, _ (the step is advanced here)."""
    _check(all(t.dtype == torch.float32 and t.is_contiguous() and t.numel() == w.numel() for t in (w, g, m, v)),
           'fused_adam_: w / g / m / v must be contiguous f32 buffers of one size')
    _check(opt_state.dtype == torch.float64 and opt_state.numel() == 8, 'fused_adam_: opt_state f64 [8]')
    be = _backend()
    be.begin_step(opt_state, None)
    be.adam(w.view(-1), g.view(-1), m.view(-1), v.view(-1), opt_state)


# ------------------------------------------------------------------------------------------------ loss heads
def _f32c(t, name, cols=None):
    _check(t.dtype == torch.float32 and t.is_cuda, f'{name}: f32 device tensor expected')
    t = t.contiguous()
    _check(cols is None or (t.dim() == 2 and t.shape[1] == cols), f'{name}: expected [rows, {cols}], got {tuple(t.shape)}')
    return t


@torch.library.custom_op('ase_hip::ppo_loss_head', mutates_args=(), device_types='cuda')
def ppo_loss_head(mu: torch.Tensor, value: torch.Tensor, actions: torch.Tensor, old_mu: torch.Tensor, old_sigma: torch.Tensor,
                  old_neglogp: torch.Tensor, advantages: torch.Tensor, old_values: torch.Tensor, returns: torch.Tensor,
                  mask: torch.Tensor, logstd: torch.Tensor, e_clip: float, critic_coef: float, bounds_coef: float,
                  clip_value: bool) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """PPO loss head of one minibatch (learning/common_agent.py:505-534,456-464; rl_games neglogp / policy_kl):
    stats = [actor_loss, critic_loss, bound_loss, entropy, clip_fraction, kl] as the reference reports them (mask: an EMPTY tensor ->
    plain means; else the AMP / ASE masked means sum(mask x) / sum(mask) of learning/amp_agent.py:316-324), d_mu = d (actor_loss +
    bounds_coef bound_loss) / d mu, d_value = d (critic_coef critic_loss) / d value.  One launch of ase_hip_ppo_head; nothing
    synchronises (the statistics are formed from the device accumulators by tensor operations)."""
    M, A = mu.shape
    _check(mu.dim() == 2 and 1 <= A <= 64, 'ppo_loss_head: mu [M, actions <= 64]')
    be, dev = _backend(), mu.device
    mu, value = _f32c(mu, 'mu'), _f32c(value.reshape(M, 1), 'value')
    masked = mask.numel() > 0
    mb = {'actions': _f32c(actions, 'actions', A), 'mu': _f32c(old_mu, 'old_mu', A), 'sigma': _f32c(old_sigma, 'old_sigma', A),
          'old_logp_actions': _f32c(old_neglogp.reshape(M, 1), 'old_neglogp'), 'advantages': _f32c(advantages.reshape(M, 1), 'advantages'),
          'old_values': _f32c(old_values.reshape(M, 1), 'old_values'), 'returns': _f32c(returns.reshape(M, 1), 'returns')}
    acc = torch.zeros(L.ACC_COUNT, dtype=torch.float64, device=dev)
    if masked:
        mb['rand_action_mask'] = _f32c(mask.reshape(M, 1).float(), 'mask')
        be.reduce_sum(mb['rand_action_mask'], M, False, acc, L.ACC_MASK_SUM)
    d_mu, d_value = torch.zeros(M, A, dtype=torch.float32, device=dev), torch.zeros(M, 1, dtype=torch.float32, device=dev)
    be.ppo_head(mu, value, mb, None, _f32c(logstd.reshape(-1), 'logstd'), d_mu, d_value, None, None, acc, M, M, A, 0, masked, False,
                False, bool(clip_value), e_clip, critic_coef, bounds_coef, 0.0, 0.0)
    den = acc[L.ACC_MASK_SUM] if masked else float(M)
    stats = torch.stack([acc[L.ACC_A_LOSS] / den, acc[L.ACC_C_LOSS] / M, acc[L.ACC_B_LOSS] / den, acc[L.ACC_ENTROPY] / den,
                         acc[L.ACC_CLIPPED] / den, acc[L.ACC_KL] / M]).float()
    return stats, d_mu, d_value


@ppo_loss_head.register_fake
def _(mu, value, actions, old_mu, old_sigma, old_neglogp, advantages, old_values, returns, mask, logstd, e_clip, critic_coef,
      bounds_coef, clip_value):
    return mu.new_empty(6), torch.empty_like(mu), mu.new_empty(mu.shape[0], 1)


@torch.library.custom_op('ase_hip::disc_loss_gp', mutates_args=(), device_types='cuda')
def disc_loss_gp(logits: torch.Tensor, grad_demo: torch.Tensor, disc_coef: float) -> tuple[torch.Tensor, torch.Tensor]:
    """Discriminator loss head (learning/amp_agent.py:442-459,481-496).  logits [3 n, 1]: rows [0, 2 n) agent + replay (target 0),
    rows [2 n, 3 n) demo (target 1); grad_demo [n, D] = d logit / d (demo observation), from autograd or the engine's chain (an EMPTY
    tensor skips the penalty).  stats = [0.5 (BCE_agent + BCE_demo), gradient penalty = mean_rows |grad|^2, agent accuracy, demo
    accuracy]; d_logits = disc_coef x d (0.5 (BCE_agent + BCE_demo)) / d logits."""
    R = logits.shape[0]
    _check(logits.dim() == 2 and logits.shape[1] == 1 and R % 3 == 0, 'disc_loss_gp: logits [3 n, 1]')
    n, dev, be = R // 3, logits.device, _backend()
    lg = _f32c(logits, 'logits')
    acc = torch.zeros(L.ACC_COUNT, dtype=torch.float64, device=dev)
    d_logit = torch.zeros(R, 1, dtype=torch.float32, device=dev)
    be.disc_head(lg, d_logit, None, acc, n, n, disc_coef)
    if grad_demo.numel():
        g = _f32c(grad_demo, 'grad_demo')
        _check(g.dim() == 2 and g.shape[0] == n, 'disc_loss_gp: grad_demo [n, D]')
        be.sqnorm(g, n, g.shape[1], acc, L.ACC_GP, scale=1.0)
    stats = torch.stack([0.5 * (acc[L.ACC_BCE_AGENT] / (2 * n) + acc[L.ACC_BCE_DEMO] / n), acc[L.ACC_GP] / n,
                         acc[L.ACC_AGENT_ACC] / (2 * n), acc[L.ACC_DEMO_ACC] / n]).float()
    return stats, d_logit


@disc_loss_gp.register_fake
def _(logits, grad_demo, disc_coef):
    return logits.new_empty(4), torch.empty_like(logits)


@torch.library.custom_op('ase_hip::enc_div_loss', mutates_args=(), device_types='cuda')
def enc_div_loss(enc_pred: torch.Tensor, enc_z: torch.Tensor, mu: torch.Tensor, mu_div: torch.Tensor, z: torch.Tensor,
                 z_new: torch.Tensor, enc_coef: float, div_coef: float,
                 div_tar: float) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """ASE's two extra loss heads (learning/ase_agent.py:413-418,445-467).  enc_pred [n, Z] = the encoder's PRE-normalisation output,
    enc_z [n, Z] the latents of those rows: enc_loss = mean(-<normalize(enc_pred), enc_z>), d_enc = enc_coef x its gradient.
    mu / mu_div [M, A] = the actor's means under the rollout's latents z [M, Z] and under fresh ones z_new [M, Z]:
    diversity loss = mean((div_tar - |clamp(mu) - clamp(mu_div)|^2 / A / (0.5 - 0.5 <z_new, z> + 1e-5))^2), d_mu / d_mu_div =
    div_coef x its gradients.  stats = [enc_loss, diversity_loss]."""
    n, Z = enc_pred.shape
    M, A = mu.shape
    _check(enc_z.shape == enc_pred.shape and Z <= 128, 'enc_div_loss: enc_pred / enc_z [n, Z <= 128]')
    _check(mu_div.shape == mu.shape and z.shape == (M, Z) and z_new.shape == (M, Z) and A <= 64, 'enc_div_loss: mu / mu_div [M, A], z / z_new [M, Z]')
    be, dev = _backend(), mu.device
    f32 = dict(dtype=torch.float32, device=dev)
    acc = torch.zeros(L.ACC_COUNT, dtype=torch.float64, device=dev)
    d_enc = torch.zeros(n, Z, **f32)
    be.enc_head(_f32c(enc_pred, 'enc_pred'), _f32c(enc_z, 'enc_z'), d_enc, None, None, acc, n, n, Z, enc_coef)
    # the diversity term rides in ase_hip_ppo_head's launch: with zero advantages, zero critic / bound coefficients its other terms vanish
    mu2 = torch.cat([_f32c(mu, 'mu'), _f32c(mu_div, 'mu_div')])
    zeros1, ones = torch.zeros(M, 1, **f32), torch.ones(M, A, **f32)
    mb = {'actions': mu2[:M], 'mu': mu2[:M], 'sigma': ones, 'old_logp_actions': zeros1, 'advantages': zeros1, 'old_values': zeros1,
          'returns': zeros1, 'ase_latents': _f32c(z, 'z')}
    d_mu2, d_v = torch.zeros(2 * M, A, **f32), torch.zeros(M, 1, **f32)
    be.ppo_head(mu2, zeros1, mb, _f32c(z_new, 'z_new'), torch.zeros(A, **f32), d_mu2, d_v, None, None, acc, M, M, A, Z, False, True,
                False, False, 0.2, 0.0, 0.0, div_coef, div_tar)
    stats = torch.stack([acc[L.ACC_ENC] / n, acc[L.ACC_DIV] / M]).float()
    return stats, d_enc, d_mu2[:M].clone(), d_mu2[M:].clone()          # (clones: a custom operator's outputs must not alias each other)


@enc_div_loss.register_fake
def _(enc_pred, enc_z, mu, mu_div, z, z_new, enc_coef, div_coef, div_tar):
    return mu.new_empty(2), torch.empty_like(enc_pred), torch.empty_like(mu), torch.empty_like(mu)
