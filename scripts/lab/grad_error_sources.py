"""Where the 16-bit POLICY-gradient error comes from (round 4 verdict, item 2): a by-ingredient budget for the actor branch.

Runs on the CPU (no GPU, no library): the actor branch of the ASE step at full width - style MLP 64 -> 512 -> 256 -> 64 (tanh),
[obs 253 | style 64] -> 1024 -> 1024 -> 512 -> mu 31, ReLU - with the PPO surrogate + bound loss of
learning/common_agent.py:372-383,505-519 on a 16384-row minibatch, in f32 autograd.  Each storage rounding the half-precision
engine performs is emulated by a straight-through rounding op and switched on ALONE, then all together:

    M    (not a rounding) ReLU masks: "all, masks of the f32 forward" runs every rounding above but takes the 0/1 derivative
         masks from the EXACT forward - what is left is the error of the arithmetic proper; the rows above it carry, in
         addition, every unit whose pre-activation changes sign under the rounding (|z| of the order of the rounding error):
         each such flip is an O(1) error of one element of dZ, a fraction f of flipped elements is a relative L2 error sqrt(f)
    W    weight shadows (W_s in the forward, W_s^T in the data-gradient launches) rounded to half
    X    inputs (normalised observations, latents) stored in half
    H    hidden activations stored in half (read again by the next layer AND by the weight-gradient launches)
    D    back-propagated gradients dZ_l stored in half under the static gradient scale S = 4096 (saturating)

against the f32 gradient on IDENTICAL inputs, per tensor, as relative L2 error - the quantity bench.py / the GPU tests report.
Two definitions of the step's inputs:

    fresh      mu_old / neglogp_old come from the HALF forward of the same weights (the engine's own inference path filled
               the rollout - bench.py's `fresh` state): the engine's importance ratio is exactly 1, the f32 oracle's is
               exp(neglogp_old - neglogp_f32) = 1 + O((a - mu) / sigma^2 * d_mu), d_mu = the half forward's error in mu
    consistent mu_old / neglogp_old come from the f32 forward: both sides see a ratio that differs from 1 only by THEIR OWN
               forward error (what a reference run on its own rollout sees)

    python scripts/lab/grad_error_sources.py [rows] > profiles/r05_grad_error_sources.txt
"""
import math
import sys

import torch

torch.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
S = 4096.0
SIGMA = math.exp(-2.9)
E_CLIP, BOUND_COEF = 0.2, 10.0


class _Round(torch.autograd.Function):          # forward: round to half; backward: identity (straight-through)
    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):      # forward: identity; backward: the gradient stored in half under the scale S
    @staticmethod
    def forward(ctx, x):
        return x

    @staticmethod
    def backward(ctx, g):
        return (g * S).clamp(-65504.0, 65504.0).half().float() / S


def rnd(x, on):
    return _Round.apply(x) if on else x


def rgrad(x, on):
    return _RoundGrad.apply(x) if on else x


def linear_init(n, k, bound=None):
    b = 1.0 / math.sqrt(k) if bound is None else bound
    return ((torch.rand(n, k) * 2 - 1) * b).requires_grad_(True), torch.zeros(n, requires_grad=True)


def build():
    P = {}
    P['style0.w'], P['style0.b'] = linear_init(512, 64)
    P['style1.w'], P['style1.b'] = linear_init(256, 512)
    P['style_dense.w'], P['style_dense.b'] = linear_init(64, 256, bound=1.0)        # ase_network_builder.py:327,335: U(-1, 1)
    P['dense0.w'], P['dense0.b'] = linear_init(1024, 317)
    P['dense1.w'], P['dense1.b'] = linear_init(1024, 1024)
    P['dense2.w'], P['dense2.b'] = linear_init(512, 1024)
    P['mu.w'], P['mu.b'] = linear_init(31, 512)
    return P


def forward(P, obs, z, W=False, X=False, H=False, D=False, masks=None, record=None):
    """masks: {layer: 0/1 tensor} - ReLU replaced by multiplication with GIVEN derivative masks (those of another forward);
    record: dict that receives this forward's own masks."""
    def lin(x, name, act):
        y = x @ rnd(P[name + '.w'], W).t() + P[name + '.b']
        if act == 'relu':
            if record is not None:
                record[name] = (y > 0).float().detach()
            y = y * masks[name] if masks is not None else torch.relu(y)
        elif act == 'tanh':
            y = torch.tanh(y)
        return y
    zz = rnd(z, X)
    h = rgrad(rnd(lin(zz, 'style0', 'relu'), H), D)
    h = rgrad(rnd(lin(h, 'style1', 'relu'), H), D)
    st = rgrad(rnd(lin(h, 'style_dense', 'tanh'), H), D)       # written straight into the latent block of the first layer's input
    x = torch.cat([rnd(obs, X), st], dim=-1)
    h = rgrad(rnd(lin(x, 'dense0', 'relu'), H), D)
    h = rgrad(rnd(lin(h, 'dense1', 'relu'), H), D)
    h = rgrad(rnd(lin(h, 'dense2', 'relu'), H), D)
    mu = lin(h, 'mu', None)                                     # f32 head output (the engine's MU buffer is f32)
    return rgrad(mu, D)                                         # ... and dMU is stored in half


def neglogp(a, mu):
    return 0.5 * (((a - mu) / SIGMA) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * a.shape[-1] + math.log(SIGMA) * a.shape[-1]


def loss_fn(mu, a, old_nlp, adv, mask):
    ratio = torch.exp(old_nlp - neglogp(a, mu))
    al = torch.max(-adv * ratio, -adv * ratio.clamp(1 - E_CLIP, 1 + E_CLIP))
    bl = ((mu - 1.0).clamp_min(0) ** 2 + (mu + 1.0).clamp_max(0) ** 2).sum(-1)
    return ((al + BOUND_COEF * bl) * mask).sum() / mask.sum(), (al * mask).sum() / mask.sum()


def grads(P, obs, z, a, old_nlp, adv, mask, **kw):
    for p in P.values():
        p.grad = None
    mu = forward(P, obs, z, **kw)
    loss, al = loss_fn(mu, a, old_nlp, adv, mask)
    loss.backward()
    return {k: p.grad.clone() for k, p in P.items()}, float(al.detach()), mu.detach()


def rel(g, ref):
    return {k: float((g[k].double() - ref[k].double()).norm() / ref[k].double().norm().clamp_min(1e-30)) for k in ref}


def main():
    P = build()
    obs = torch.randn(M, 253).clamp(-5, 5)
    z = torch.nn.functional.normalize(torch.randn(M, 64), dim=-1)
    adv = torch.randn(M)
    mask = (torch.rand(M) < 0.8).float()
    adv = (adv - (adv * mask).sum() / mask.sum())
    eps = torch.randn(M, 31)
    with torch.no_grad():
        mu32 = forward(P, obs, z)
        mu16 = forward(P, obs, z, W=True, X=True, H=True)
    d_mu = (mu16 - mu32)
    print(f'rows {M}; |mu| rms {float(mu32.pow(2).mean().sqrt()):.3f}; half-forward error in mu: rms {float(d_mu.pow(2).mean().sqrt()):.2e} '
          f'(sigma = {SIGMA:.4f}: d_neglogp rms = {float(((eps / SIGMA) * d_mu).sum(-1).pow(2).mean().sqrt()):.2e})')
    ING = [('W', dict(W=True)), ('X', dict(X=True)), ('H', dict(H=True)), ('D', dict(D=True)), ('W+X+H (whole forward)', dict(W=True, X=True, H=True)),
           ('all', dict(W=True, X=True, H=True, D=True))]
    show = ['dense0.w', 'dense1.w', 'dense2.w', 'mu.w', 'style0.w', 'style1.w', 'style_dense.w', 'dense0.b']
    for state, mu_old in (('fresh (mu_old from the HALF forward: bench.py / tests)', mu16), ('consistent (mu_old from the f32 forward)', mu32)):
        a = mu_old + SIGMA * eps
        old_nlp = neglogp(a, mu_old)
        ref, al_ref, _ = grads(P, obs, z, a, old_nlp, adv, mask)
        print(f'\n== {state}: actor_loss(f32) = {al_ref:+.6f}')
        print(f'{"ingredient":28s} ' + ' '.join(f'{k:>13s}' for k in show) + f' {"median(all)":>12s} {"actor_loss true rel":>20s}')
        for name, kw in ING:
            g, al, _ = grads(P, obs, z, a, old_nlp, adv, mask, **kw)
            r = rel(g, ref)
            med = sorted(r.values())[len(r) // 2]
            print(f'{name:28s} ' + ' '.join(f'{r[k]:13.2e}' for k in show) + f' {med:12.2e} {abs(al - al_ref) / abs(al_ref):20.2e}')
        # every rounding on, but the ReLU derivative masks of the EXACT forward (and the reverse: exact arithmetic, the half
        # forward's masks): separates the arithmetic error from the mask flips
        rec32, rec16 = {}, {}
        with torch.no_grad():
            forward(P, obs, z, record=rec32)
            forward(P, obs, z, W=True, X=True, H=True, record=rec16)
        flips = {k: float((rec32[k] != rec16[k]).float().mean()) for k in rec32}
        for name, kw in (('all, masks of f32 forward', dict(W=True, X=True, H=True, D=True, masks=rec32)),
                         ('f32 arithmetic, half masks', dict(masks=rec16))):
            g, al, _ = grads(P, obs, z, a, old_nlp, adv, mask, **kw)
            r = rel(g, ref)
            med = sorted(r.values())[len(r) // 2]
            print(f'{name:28s} ' + ' '.join(f'{r[k]:13.2e}' for k in show) + f' {med:12.2e} {abs(al - al_ref) / abs(al_ref):20.2e}')
        print('   flipped ReLU derivative masks (half forward vs f32 forward), fraction of elements per layer: ' +
              ', '.join(f'{k} {v:.1e} (sqrt {math.sqrt(v):.1e})' for k, v in flips.items()))
        # what a more exact MU alone would buy: the half engine's gradient with the importance ratio evaluated on the f32 mu
        # (forward rounding stays in the activations the backward reads)
        for p in P.values():
            p.grad = None
        mu_h = forward(P, obs, z, W=True, X=True, H=True, D=True)
        mu_mix = mu_h + (mu32 - mu_h).detach()
        loss, al = loss_fn(mu_mix, a, old_nlp, adv, mask)
        loss.backward()
        r = rel({k: p.grad.clone() for k, p in P.items()}, ref)
        med = sorted(r.values())[len(r) // 2]
        print(f'{"all, but mu VALUE exact":28s} ' + ' '.join(f'{r[k]:13.2e}' for k in show) + f' {med:12.2e} {abs(float(al) - al_ref) / abs(al_ref):20.2e}')


if __name__ == '__main__':
    main()
