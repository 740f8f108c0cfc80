"""CPU emulation of the libase_hip C-ABI operations — TEST INFRASTRUCTURE ONLY.

Same method names and argument conventions as ``ase_amd.backend.HipBackend`` so that the engine's
HOST logic (buffer layout, launch sequence, analytic backward, index maps) can be checked against
the oracle on a machine without a GPU, and so that the GPU tests have a per-op reference with the
SAME semantics as the kernels (including padding, concat-column maps and the analytic — not
autograd — head gradients).  The product never imports this file.
"""
import math

import torch

from ase_amd import lib as L


def _dyn(t):
    """Factor of a `*_dev` / dyn argument (a scale record {factor, overflow count} on the device - include/ase_hip.h, ABI 7 - or a
    bare f32 factor), 1 when absent."""
    return 1.0 if t is None else float(t.reshape(-1)[0])


def _overflowed(x):
    """csrc/common.h ovf_hit1: a STORED element that is not finite or (IEEE half, whose conversions saturate) sits at +-65504."""
    v = x.float()
    bad = ~torch.isfinite(v)
    if x.dtype == torch.float16:
        bad |= v.abs() >= 65504.0
    return bool(bad.any())


def _report(rec, *stored):
    """What a launch that was given a scale record does with what it stored: count += (something positive) on an overflow."""
    if rec is not None and any(_overflowed(x) for x in stored):
        assert rec.numel() >= 2, "a launch that may report needs a {factor, count} record"
        rec.reshape(-1)[1] += 1.0


def _rows(idx, remap, M):
    p = torch.arange(M) if idx is None else idx[:M].long()
    if remap[0] > 0:
        H, N = remap
        env = p // H
        t = p - env * H
        p = t * N + env
    return p


def _store(v, dtype):
    """f32 -> storage type as the kernels convert: round to nearest even; half saturates at its largest finite value."""
    if dtype == torch.float16:
        v = v.clamp(-65504.0, 65504.0)
    return v.to(dtype)


def _act_fns(act):
    """(f, f', f'') of an activation as functions of the pre-activation (csrc/act.h)."""
    import torch.nn.functional as F
    sg = torch.sigmoid
    phi = lambda z: torch.exp(-0.5 * z * z) * 0.3989422804014327
    Phi = lambda z: 0.5 * (1 + torch.erf(z * 0.7071067811865476))
    lam, al = 1.0507009873554805, 1.6732632423543772
    return {
        L.ACT_NONE: (lambda z: z, lambda z: torch.ones_like(z), lambda z: torch.zeros_like(z)),
        L.ACT_RELU: (torch.relu, lambda z: (z > 0).to(z.dtype), lambda z: torch.zeros_like(z)),
        L.ACT_TANH: (torch.tanh, lambda z: 1 - torch.tanh(z) ** 2, lambda z: -2 * torch.tanh(z) * (1 - torch.tanh(z) ** 2)),
        L.ACT_SILU: (F.silu, lambda z: sg(z) * (1 + z * (1 - sg(z))), lambda z: sg(z) * (1 - sg(z)) * (2 + z * (1 - 2 * sg(z)))),
        L.ACT_ELU: (F.elu, lambda z: torch.where(z > 0, torch.ones_like(z), torch.exp(z)),
                    lambda z: torch.where(z > 0, torch.zeros_like(z), torch.exp(z))),
        L.ACT_GELU: (F.gelu, lambda z: Phi(z) + z * phi(z), lambda z: phi(z) * (2 - z * z)),
        L.ACT_SIGMOID: (sg, lambda z: sg(z) * (1 - sg(z)), lambda z: sg(z) * (1 - sg(z)) * (1 - 2 * sg(z))),
        L.ACT_SELU: (F.selu, lambda z: lam * torch.where(z > 0, torch.ones_like(z), al * torch.exp(z)),
                     lambda z: lam * torch.where(z > 0, torch.zeros_like(z), al * torch.exp(z))),
        L.ACT_SOFTPLUS: (F.softplus, sg, lambda z: sg(z) * (1 - sg(z))),
    }[act]


def _twin_factors(act, t):
    """(act', act'' / act') from a layer's twin: its output for ReLU / tanh, its pre-activation otherwise."""
    if act == L.ACT_RELU:
        return (t > 0).float(), torch.zeros_like(t)
    if act == L.ACT_TANH:
        d1 = 1 - t * t
        d2 = -2 * t * d1
    else:
        _, f1, f2 = _act_fns(act)
        d1, d2 = f1(t), f2(t)
    return d1, torch.where(d1 != 0, d2 / d1, torch.zeros_like(d1))


def half_split(x, e):
    """x (f32) -> (hi, lo) as the ASE_F32H3 kernels form them: sx = x * 2^e, hi = half(sx) saturating at +-65504, lo = half(sx - hi)
    (csrc/gemm_nt_kernels.h split_f16), both returned as f64 values."""
    sx = x.float() * (2.0 ** e)
    hi = sx.clamp(-65504.0, 65504.0).half()
    lo = (sx - hi.float()).clamp(-65504.0, 65504.0).half()
    return hi.double(), lo.double()


def unpack_split(buf, rows, cols, e):
    """A packed half-split shadow (ase_hip_refresh_shadow with ASE_F32H3: per group of 8 elements 8 hi halves then 8 lo halves,
    in a float32-typed buffer) -> (hi + lo) / 2^e as f64 [rows, cols]."""
    raw = buf.contiguous().view(torch.float16).view(buf.shape[0], -1, 2, 8)        # [row, group, hi|lo, 8]
    v = (raw[:, :, 0, :].double() + raw[:, :, 1, :].double()).reshape(buf.shape[0], -1)
    return v[:rows, :cols] * 2.0 ** -e


def x3_half_product(a, b, ea, eb):
    """a @ b^T as the three-product form hi*hi + hi*lo + lo*hi of the scaled half splits, exact accumulation (f64), scale undone."""
    ah, al = half_split(a, ea)
    bh, bl = half_split(b, eb)
    v = ah @ bh.t() + ah @ bl.t() + al @ bh.t()
    return (v * 2.0 ** -(ea + eb)).float()


class EmuBackend:
    name = "emu"

    def __init__(self, group_all=False):
        self.rng = torch.Generator().manual_seed(99)
        self.group_all = group_all        # queue EVERY weight gradient for the grouped launch (tiny test shapes)
        self.grouped_launches = 0

    def zero_(self, t):
        t.zero_()

    def copy_(self, dst, src):
        dst.copy_(src)

    def host_call(self, fn):
        fn()

    def mark(self):
        return 0

    def wait(self, ev):
        pass

    # ------------------------------------------------------------------ GEMMs
    def gemm_nt(self, A, B, Cm, M, N, K, bias=None, aux=None, aux_mode=L.AUX_NONE, colsum=None, colsum_n=0,
                act=L.ACT_NONE, alpha=1.0, aux_split=0, aux_delta=0, mask_out=None, x3_exps=None, alpha_dev=None):
        alpha = alpha * _dyn(alpha_dev)
        if aux is not None and aux_split > 0:
            rows = torch.arange(M)
            rows = torch.where(rows >= aux_split, rows - aux_delta, rows)
            aux = aux[rows]
        a = A[:M, :K].float()
        b = B[:N, :K].float()
        if x3_exps is not None and getattr(self, 'x3', None) == 'f16':
            v = alpha * x3_half_product(a, b, *x3_exps)
        else:
            v = alpha * (a @ b.t())
        if bias is not None:
            v = v + bias[:N]
        z_pre = v
        if act == L.ACT_RELU:
            v = torch.relu(v)
        elif act == L.ACT_TANH:
            v = torch.tanh(v)
        elif act != L.ACT_NONE:
            v = _act_fns(act)[0](v)
        if (aux_mode & 0xFF) == L.AUX_PREACT:
            v = v * _act_fns(aux_mode >> 8)[1](aux[:M, :N].float())
        elif aux_mode == L.AUX_RELU_MASK:
            v = v * (aux[:M, :N].float() > 0)
        elif aux_mode == L.AUX_RELU_BITS:         # uint32 words, 32 columns each (stored as int32)
            w = aux[:M, :(N + 31) // 32].to(torch.int64) & 0xFFFFFFFF
            bits = (w.unsqueeze(-1) >> torch.arange(32)) & 1
            v = v * bits.reshape(M, -1)[:, :N].to(v.dtype)
        elif aux_mode == L.AUX_TANH_GRAD:
            x = aux[:M, :N].float()
            v = v * (1 - x * x)
        out = _store(v, Cm.dtype)
        Cm[:M, :N] = out
        _report(alpha_dev, out)
        if mask_out is not None and act >= L.ACT_SILU:        # twin of a smooth activation: the pre-activation itself
            mask_out[:M, :N] = _store(z_pre, mask_out.dtype)
            _report(alpha_dev, mask_out[:M, :N])
        elif mask_out is not None:
            assert N % 32 == 0
            b = (out.float() > 0).to(torch.int64).reshape(M, N // 32, 32)
            w = (b << torch.arange(32)).sum(-1)
            w = torch.where(w >= 2 ** 31, w - 2 ** 32, w)
            mask_out[:M, :N // 32] = w.to(torch.int32)
        if colsum is not None and colsum_n > 0:
            colsum[:colsum_n] += out.float().sum(0)[:colsum_n]

    def gemm_tn(self, A, B, G, M, N, K, n_real, k_real, split_src, split_dst, alpha=1.0, gbias=None, bias_rows=0, alpha_dev=None):
        alpha = alpha * _dyn(alpha_dev)
        if gbias is not None:
            br = bias_rows if bias_rows > 0 else M
            gbias[:n_real] += alpha * A[:br, :n_real].float().sum(0)
        full = alpha * (A[:M, :N].float().t() @ B[:M, :K].float())       # [N, K] padded layout
        gap = split_dst - split_src
        cols = list(range(split_src)) + [k for k in range(split_dst, K) if k - gap < k_real]
        G[:n_real, :] += full[:n_real][:, cols]

    def grouped_tn_ok(self, dtype, M, n_real, K, bias_rows):
        if self.group_all:
            return True
        return M % 64 == 0 and bias_rows % 64 == 0 and n_real >= 128 and K >= 128      # any dtype: exercises the deferral on CPU

    def make_tn_plan(self, problems, target_wg=0, alpha_dev=None):
        return {'keep': problems, 'alpha_dev': alpha_dev}

    def gemm_tn_grouped(self, plan):
        self.grouped_launches += 1
        for (A, B, G, gb, br, M, N, K, nr, kr, ss, sd, alpha) in plan['keep']:
            self.gemm_tn(A, B, G, M, N, K, nr, kr, ss, sd, alpha=alpha, gbias=gb, bias_rows=br, alpha_dev=plan.get('alpha_dev'))

    def refresh_shadow(self, W, Ws, Wts, split_src, split_dst, x3_exp=None):
        # (x3_exp: the HIP backend packs half splits there; the emulator keeps plain f32 shadows - its x3 products are formed
        #  from the f32 values by x3_half_product)
        n, k = W.shape
        gap = split_dst - split_src
        kd = torch.tensor([j if j < split_src else j + gap for j in range(k)])
        if Ws is not None:
            Ws[:n, kd] = W.to(Ws.dtype)
        if Wts is not None:
            Wts[kd, :n] = W.t().to(Wts.dtype)

    def refresh_shadow_multi(self, desc, items, dtype):
        for W, ws, wts, ss, sd, b, bs in items:       # the pointer table `desc` describes exactly these tensors
            self.refresh_shadow(W, ws, wts, ss, sd)
            bs[:b.numel()] = b

    def apply_multi(self, desc, items, dtype, opt_state, acc):
        for (W, ws, wts, ss, sd, b, bs, gW, mW, vW, gb, mb, vb, coef, slot_a, slot_b) in items:
            if opt_state is not None:
                if slot_a >= 0:
                    s2 = (W.double() ** 2).sum()
                    acc[slot_a] += s2
                    if slot_b >= 0:
                        acc[slot_b] += s2
                if coef != 0:
                    self.axpy(gW, W, coef)
                self.adam(W, gW, mW, vW, opt_state)
                self.adam(b, gb, mb, vb, opt_state)
            self.refresh_shadow(W, ws, wts, ss, sd)
            bs[:b.numel()] = b

    def gather_multi(self, desc, items, idx, remap, M):
        for src, D, dst in items:
            self.gather_rows(src, D, idx, remap, M, dst)

    # ------------------------------------------------------------------ normaliser / gather
    def rms_moments(self, src, D, idx, remap, M, state, sums):
        x = src[_rows(idx, remap, M), :D]
        shift = state[:D].float()
        d = (x - shift).double()
        sums[:D] += d.sum(0)
        sums[D:2 * D] += (d * d).sum(0)

    def rms_moments_multi(self, streams, D, M, state, sums_list):
        for (src, idx, remap), sums in zip(streams, sums_list):
            self.rms_moments(src, D, idx, remap, M, state, sums)

    def rms_normalize_multi(self, streams, D, M, means, stds, outs):
        for (src, idx, remap), mean, std, out in zip(streams, means, stds, outs):
            self.rms_normalize(src, D, idx, remap, M, mean, std, [out])

    def rms_finalize(self, state, D, sums, count, n_streams, mean_out, std_out):
        mean, var, cnt = state[:D].clone(), state[D:2 * D].clone(), state[2 * D].clone()
        shift = mean.float().double()
        mo, so = mean_out.view(-1, D), std_out.view(-1, D)
        if n_streams == 0:
            mo[0] = mean.float()
            so[0] = torch.sqrt(var.float() + 1e-5)
        sm = sums.view(-1, 2 * D) if n_streams > 0 else None
        for s in range(n_streams):
            n = float(count)
            s1, s2 = sm[s, :D], sm[s, D:]
            bm = (shift + s1 / n).float().double()
            bv = ((s2 - s1 * s1 / n) / (n - 1.0)).float().double()
            delta = bm - mean
            tot = cnt + n
            new_mean = mean + delta * n / tot
            m2 = var * cnt + bv * n + delta * delta * cnt * n / tot
            mean, var, cnt = new_mean, m2 / tot, tot
            mo[s] = mean.float()
            so[s] = torch.sqrt(var.float() + 1e-5)
        state[:D], state[D:2 * D], state[2 * D] = mean, var, cnt

    def rms_normalize(self, src, D, idx, remap, M, mean, std, outs):
        x = src[_rows(idx, remap, M), :D]
        y = torch.clamp((x - mean[:D]) / std[:D], -5.0, 5.0)
        for o in outs:
            if o is not None:
                o[:M, :D] = y.to(o.dtype)

    def rms_unnormalize(self, state, x, y):
        y.copy_(torch.sqrt(state[1].float() + 1e-5) * torch.clamp(x, -5.0, 5.0) + state[0].float())

    def gather_rows(self, src, D, idx, remap, M, dst):
        dst[:M, :D] = src[_rows(idx, remap, M), :D].to(dst.dtype)

    # ------------------------------------------------------------------ heads
    def reduce_sum(self, x, n, square, acc, slot):
        v = x.reshape(-1)[:n].double()
        acc[slot] += (v * v).sum() if square else v.sum()

    def ppo_head(self, mu, value, mb, new_z, logstd, d_mu, d_value, db_mu, db_value, acc, M, m_global, act_dim,
                 z_dim, masked, div_on, mu_tanh, clip_value, e_clip, critic_coef, bounds_coef, div_coef, div_tar,
                 mu_out=None, grad_scale=1.0, dyn=None):
        D, gs = act_dim, float(grad_scale) * _dyn(dyn)
        raw = mu[:M, :D]
        m = torch.tanh(raw) if mu_tanh else raw
        a, omu, osg = mb['actions'], mb['mu'], mb['sigma']
        ls = logstd[:D]
        sg = torch.exp(ls)
        d = (a - m) / sg
        nlp = 0.5 * (d * d).sum(-1) + 0.5 * math.log(2 * math.pi) * D + ls.sum()
        ratio = torch.exp(mb['old_logp_actions'].view(-1) - nlp)
        adv = mb['advantages'].view(-1)
        rc = torch.clamp(ratio, 1 - e_clip, 1 + e_clip)
        s1, s2 = -adv * ratio, -adv * rc
        a_loss = torch.max(s1, s2)
        g = torch.where(ratio == rc, -adv, torch.where(s1 > s2, -adv, torch.where(s1 == s2, -0.5 * adv, torch.zeros_like(adv))))
        S = float(acc[L.ACC_MASK_SUM]) if masked else float(m_global)
        mk = mb['rand_action_mask'].view(-1) if masked else torch.ones(M)
        w = mk / S
        bh, bl = torch.clamp_min(m - 1, 0), torch.clamp_max(m + 1, 0)
        b_row = (bh * bh + bl * bl).sum(-1)
        ent_row = (0.5 + 0.5 * math.log(2 * math.pi) + ls).sum().expand(M)
        kl_row = (torch.log(osg / sg + 1e-5) + (sg * sg + (omu - m) ** 2) / (2 * (osg * osg + 1e-5)) - 0.5).sum(-1)
        gm = (w * g * ratio).unsqueeze(-1) * d / sg + bounds_coef * w.unsqueeze(-1) * 2 * (bh + bl)
        div_row = torch.zeros(M)
        gm2 = None
        if div_on:
            raw2 = mu[M:2 * M, :D]
            m2 = torch.tanh(raw2) if mu_tanh else raw2
            cm, cm2 = torch.clamp(m, -1, 1), torch.clamp(m2, -1, 1)
            diff = cm - cm2
            a_diff = (diff * diff).sum(-1) / D
            zz = (new_z[:M] * mb['ase_latents']).sum(-1)
            inv = 1.0 / (0.5 - 0.5 * zz + 1e-5)
            bonus = a_diff * inv
            div_row = (div_tar - bonus) ** 2
            dl = div_coef * w * 2 * (bonus - div_tar)
            db = (dl * inv).unsqueeze(-1) * 2 * diff / D
            gm = gm + db * ((m >= -1) & (m <= 1))
            gm2 = -db * ((m2 >= -1) & (m2 <= 1))
            if mu_tanh:
                gm2 = gm2 * (1 - m2 * m2)
        if mu_tanh:
            gm = gm * (1 - m * m)
        o1 = _store(gs * gm, d_mu.dtype)
        d_mu[:M, :D] = o1
        dbm = o1.float().sum(0) / gs
        if div_on:
            o2 = _store(gs * gm2, d_mu.dtype)
            d_mu[M:2 * M, :D] = o2
            dbm = dbm + o2.float().sum(0) / gs
        v = value[:M, 0]
        R = mb['returns'].view(-1)
        if clip_value:
            ov = mb['old_values'].view(-1)
            dlt = v - ov
            vpc = ov + torch.clamp(dlt, -e_clip, e_clip)
            l1, l2 = (v - R) ** 2, (vpc - R) ** 2
            c = torch.max(l1, l2)
            g1, g2 = 2 * (v - R), 2 * (vpc - R) * ((dlt >= -e_clip) & (dlt <= e_clip))
            dv = torch.where(l1 > l2, g1, torch.where(l1 < l2, g2, 0.5 * (g1 + g2)))
        else:
            c = (R - v) ** 2
            dv = 2 * (v - R)
        ov_ = _store(gs * (critic_coef * dv / m_global), d_value.dtype)
        d_value[:M, 0] = ov_
        _report(dyn, ov_, d_mu[:2 * M if div_on else M, :D])
        if db_mu is not None:
            db_mu[:D] += dbm
            if db_value is not None:
                db_value[0] += ov_.float().sum() / gs
        if mu_out is not None:
            mu_out[:M, :D] = m
        acc[L.ACC_A_LOSS] += (mk * a_loss).double().sum()
        acc[L.ACC_B_LOSS] += (mk * b_row).double().sum()
        acc[L.ACC_ENTROPY] += (mk * ent_row).double().sum()
        acc[L.ACC_CLIPPED] += (mk * ((ratio - 1).abs() > e_clip)).double().sum()
        acc[L.ACC_C_LOSS] += c.double().sum()
        acc[L.ACC_KL] += kl_row.double().sum()
        if div_on:
            acc[L.ACC_DIV] += (mk * div_row).double().sum()

    def disc_head(self, logit, d_logit, db_logit, acc, amb, amb_global, disc_coef, grad_scale=1.0, dyn=None):
        grad_scale = grad_scale * _dyn(dyn)
        l = logit[:3 * amb, 0]
        la, ld = l[:2 * amb], l[2 * amb:]
        sp = lambda x: torch.clamp_min(x, 0) + torch.log1p(torch.exp(-x.abs()))
        acc[L.ACC_BCE_AGENT] += sp(la).double().sum()
        acc[L.ACC_BCE_DEMO] += sp(-ld).double().sum()
        acc[L.ACC_AGENT_ACC] += (la < 0).double().sum()
        acc[L.ACC_DEMO_ACC] += (ld > 0).double().sum()
        ga = disc_coef * 0.5 * torch.sigmoid(la) / (2.0 * amb_global)
        gd = -disc_coef * 0.5 * torch.sigmoid(-ld) / amb_global
        o = _store(grad_scale * torch.cat([ga, gd]), d_logit.dtype)
        d_logit[:3 * amb, 0] = o
        _report(dyn, o)
        if db_logit is not None:
            db_logit[0] += o.float().sum() / grad_scale

    def enc_head(self, e, z, d_e, db_enc, enc_out, acc, amb, amb_global, z_dim, enc_coef, grad_scale=1.0, dyn=None):
        grad_scale = grad_scale * _dyn(dyn)
        ev, zv = e[:amb, :z_dim], z[:amb, :z_dim]
        nrm = ev.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        h = ev / nrm
        dot = (h * zv).sum(-1, keepdim=True)
        o = _store(grad_scale * (-(enc_coef / amb_global) * (zv - h * dot) / nrm), d_e.dtype)
        d_e[:amb, :z_dim] = o
        _report(dyn, o)
        if db_enc is not None:
            db_enc[:z_dim] += o.float().sum(0) / grad_scale
        if enc_out is not None:
            enc_out[:amb, :z_dim] = h
        acc[L.ACC_ENC] += (-dot).double().sum()

    def enc_gp_seed(self, e, z, u, rows, z_dim, scale=1.0):
        ev, zv = e[:rows, :z_dim], z[:rows, :z_dim]
        nrm = ev.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        h = ev / nrm
        a = (h * zv).sum(-1, keepdim=True)
        u[:rows, :z_dim] = (-scale * (zv - h * a) / nrm).to(u.dtype)

    def enc_gp_back(self, e, z, du, d_e, db_enc, rows, z_dim, grad_scale=1.0, dyn=None):
        grad_scale = grad_scale * _dyn(dyn)
        ev, zv, r = e[:rows, :z_dim], z[:rows, :z_dim], du[:rows, :z_dim]
        nrm = ev.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        h = ev / nrm
        a = (h * zv).sum(-1, keepdim=True)
        hr, zr = (h * r).sum(-1, keepdim=True), (zv * r).sum(-1, keepdim=True)
        jr = (zv * hr + h * zr + a * r - 3 * a * h * hr) / (nrm * nrm)
        old = d_e[:rows, :z_dim].float().clone()
        new = _store(old + grad_scale * jr, d_e.dtype)
        d_e[:rows, :z_dim] = new
        _report(dyn, new)
        if db_enc is not None:
            db_enc[:z_dim] += (new.float() - old).sum(0) / grad_scale

    def gp_seed(self, h, w, g, rows, width, scale=1.0, act=L.ACT_RELU):
        d1, _ = _twin_factors(act, h[:rows, :width].float())
        g[:rows, :width] = _store(scale * w[:width] * d1, g.dtype)

    def gp_second(self, twin, g, dg, dz, rows, width, act):
        # (act'' / act') * (g / act') * dg, as ase_hip_gp_second: act'^2 underflows for saturated units while act' does not
        d1, c = _twin_factors(act, twin[:rows, :width].float())
        e = c * torch.where(d1 != 0, g[:rows, :width].float() / d1, torch.zeros_like(d1)) * dg[:rows, :width].float()
        gin, dgin = g[:rows, :width].float(), dg[:rows, :width].float()
        e = torch.where(torch.isfinite(e) | ~torch.isfinite(gin) | ~torch.isfinite(dgin), e, torch.zeros_like(e))     # (inputs' NaN / inf pass)
        dz[:rows, :width] = _store(dz[:rows, :width].float() + e, dz.dtype)

    def colsum(self, x, rows, cols, out, scale=1.0):
        out[:cols] += scale * x[:rows, :cols].float().sum(0)

    def sqnorm(self, x, rows, cols, acc, slot, scale=1.0, dyn=None):
        acc[slot] += scale * _dyn(dyn) * (x[:rows, :cols].double() ** 2).sum()

    def finalize_scalars(self, acc, out, m_global, amb_global, masked, has_disc, has_enc, has_div, c, opt_state=None, kl_threshold=0.0):
        a = acc.tolist()
        S = a[L.ACC_MASK_SUM]
        den = S if masked else float(m_global)
        al, bl, ent, cf = a[L.ACC_A_LOSS] / den, a[L.ACC_B_LOSS] / den, a[L.ACC_ENTROPY] / den, a[L.ACC_CLIPPED] / den
        cl, kl = a[L.ACC_C_LOSS] / m_global, a[L.ACC_KL] / m_global
        lr_used = 0.0
        if opt_state is not None:                 # rl_games AdaptiveScheduler (schedulers.py), 'legacy' schedule
            lr = cur = lr_used = float(opt_state[1])
            if kl > 2.0 * kl_threshold:
                lr = max(cur / 1.5, 1e-6)
            if kl < 0.5 * kl_threshold:
                lr = min(cur * 1.5, 1e-2)
            opt_state[1] = lr
        loss = al + c['critic_coef'] * cl - c['entropy_coef'] * ent + float(c.get('bounds_loss_coef') or 0.0) * bl
        out.zero_()
        out[L.RES_A_LOSS], out[L.RES_C_LOSS], out[L.RES_B_LOSS] = al, cl, bl
        out[L.RES_ENTROPY], out[L.RES_CLIP_FRAC], out[L.RES_KL], out[L.RES_MASK_SUM] = ent, cf, kl, S
        out[L.RES_LR] = lr_used                   # the rate this step was taken with (0 with a constant schedule)
        if has_disc:
            amb = float(amb_global)
            bce = 0.5 * (a[L.ACC_BCE_AGENT] / (2 * amb) + a[L.ACC_BCE_DEMO] / amb)
            gp = a[L.ACC_GP] / amb
            dl = bce + c['disc_logit_reg'] * a[L.ACC_LOGIT_W2] + c['disc_grad_penalty'] * gp + \
                c['disc_weight_decay'] * a[L.ACC_DISC_W2]
            loss += c['disc_coef'] * dl
            out[L.RES_DISC_LOSS], out[L.RES_DISC_GP], out[L.RES_DISC_LOGIT_LOSS] = dl, gp, a[L.ACC_LOGIT_W2]
            out[L.RES_DISC_AGENT_ACC] = a[L.ACC_AGENT_ACC] / (2 * amb)
            out[L.RES_DISC_DEMO_ACC] = a[L.ACC_DEMO_ACC] / amb
        if has_enc:
            egp = a[L.ACC_ENC_GP] / amb_global
            el = a[L.ACC_ENC] / amb_global + c.get('enc_weight_decay', 0) * a[L.ACC_ENC_W2] + c.get('enc_grad_penalty', 0) * egp
            loss += c['enc_coef'] * el
            out[L.RES_ENC_LOSS] = el
            out[L.RES_ENC_GP] = egp
        if has_div:
            dv = a[L.ACC_DIV] / S
            loss += c['amp_diversity_bonus'] * dv
            out[L.RES_DIV_LOSS] = dv
        out[L.RES_LOSS] = loss

    # ------------------------------------------------------------------ optimizer
    def begin_step(self, opt_state, acc, zero2=None, rng_bump=None):
        if acc is not None:
            acc.zero_()
        if zero2 is not None:
            zero2.zero_()
        if rng_bump is not None:
            rng_bump[1] += 1
        if opt_state is not None:
            opt_state[0] += 1
            opt_state[5] = 1.0 - float(opt_state[2]) ** float(opt_state[0])
            opt_state[6] = 1.0 - float(opt_state[3]) ** float(opt_state[0])

    def adam(self, w, g, m, v, st):
        s = st.tolist()
        b1m, b2, b2m, eps = torch.tensor(1.0 - s[2]).float(), torch.tensor(s[3]).float(), torch.tensor(1.0 - s[3]).float(), s[4]
        step_size = torch.tensor(s[1] / s[5]).float()
        bc2s = torch.tensor(math.sqrt(s[6])).float()
        m.add_(b1m * (g - m))
        v.mul_(b2).add_(b2m * (g * g))
        denom = v.sqrt() / bc2s + eps
        w.sub_(step_size * (m / denom))

    def clip_scale(self, g, acc, slot, max_norm):
        total = float(acc[slot]) ** 0.5
        g.mul_(min(max_norm / (total + 1e-6), 1.0))

    def axpy(self, g, w, c):
        g.add_(c * w)

    def scaler_check(self, buf, scaler):
        """csrc/scaler.hip: not finite, or (half storage, whose conversions saturate) at +-65504."""
        if _overflowed(buf):
            scaler[0] += 1.0

    def scaler_check_multi(self, bufs, scaler, table=None):
        for t in bufs:
            self.scaler_check(t, scaler)

    def scaler_fold(self, scaler, scale_tab):
        scaler[0] += float(scale_tab[1::2].sum())
        scale_tab[1::2] = 0.0

    def scaler_step(self, scaler, opt_state, opt_eff, grads, scale_tab=None):
        found = float(scaler[0]) != 0.0 or (scale_tab is not None and float(scale_tab[1::2].sum()) != 0.0)
        if found:
            grads.zero_()
            opt_state[0] -= 1.0
            opt_eff.copy_(opt_state)
            opt_eff[1] = 0.0
            opt_eff[2] = 1.0
            opt_eff[3] = 1.0
            opt_eff[5] = 1.0
            opt_eff[6] = 1.0
            scaler[1] += 1.0
            scaler[2] = 0.0
        else:
            opt_eff.copy_(opt_state)
            scaler[2] += 1.0
        scaler[3] += 1.0
        scaler[0] = 0.0
        if scale_tab is not None:          # GradScaler.update(), every step (csrc/scaler.hip scaler_book_kernel)
            s = float(scaler[4])
            if found:
                s *= float(scaler[6])
            elif float(scaler[2]) >= float(scaler[7]):
                s *= float(scaler[5])
                scaler[2] = 0.0
            scaler[4] = s
            scale_tab[0], scale_tab[2], scale_tab[4], scale_tab[6] = s, 1.0 / s, 1.0 / (s * s), 1.0
            scale_tab[1::2] = 0.0

    # ------------------------------------------------------------------ rollout tail
    def disc_reward(self, logit, r, n, scale):
        l = logit.reshape(-1, logit.shape[-1])[:n, 0]
        prob = 1 / (1 + torch.exp(-l))
        r.view(-1)[:n] = -torch.log(torch.clamp_min(1 - prob, 0.0001)) * scale

    def enc_reward(self, e, z, r, n, z_dim, scale):
        ev, zv = e[:n, :z_dim], z.reshape(-1, z.shape[-1])[:n, :z_dim]
        h = ev / ev.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        r.view(-1)[:n] = torch.clamp_min((h * zv).sum(-1), 0) * scale

    def gae(self, dones, values, next_values, r_task, r_disc, r_enc, w_task, w_disc, w_enc, gamma, tau, advs, returns,
            H, N):
        d = dones.view(H, N).float()
        v, nv = values.view(H, N), next_values.view(H, N)
        r = w_task * r_task.view(H, N)
        if r_disc is not None:
            r = r + w_disc * r_disc.view(H, N)
        if r_enc is not None:
            r = r + w_enc * r_enc.view(H, N)
        last = torch.zeros(N)
        A, Rt = advs.view(H, N), returns.view(H, N)
        gt = torch.tensor(gamma * tau).float()
        for t in reversed(range(H)):
            delta = r[t] + torch.tensor(gamma).float() * nv[t] - v[t]
            last = delta + gt * (1 - d[t]) * last
            A[t] = last
            Rt[t] = last + v[t]

    def adv_norm(self, returns, values, mask, adv, acc3, n, normalize, phase):
        a = (returns.view(-1) - values.view(-1))[:n]
        m = mask.view(-1)[:n] if mask is not None else torch.ones(n)
        if phase == 0:
            am = (a * m).double()
            acc3[0] += m.double().sum()
            acc3[1] += am.sum()
            acc3[2] += (am * am).sum()
            return
        if normalize:
            S, mu = float(acc3[0]), float(acc3[1]) / float(acc3[0])
            min_sqr = float(acc3[2]) / S - mu * mu
            denom = torch.tensor(math.sqrt(min_sqr * S / (S - 1))).float() + 1e-8
            adv.view(-1)[:n] = (a - torch.tensor(mu).float()) / denom
        else:
            adv.view(-1)[:n] = a

    def ring_store(self, src, D, idx, remap, n, dst, size, head):
        q = (head + torch.arange(n)) % size
        dst[q, :D] = src[_rows(idx, remap, n), :D]

    def normalize_rows(self, x, y, n, dim):
        v = x[:n, :dim].float()
        y[:n, :dim] = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)

    def sample_actions(self, mu, logstd, rand_probs, rng_state, mu_out, sigma_out, actions, neglogp, rand_mask, n, act_dim,
                       mu_tanh=False):
        g = torch.Generator().manual_seed(int(rng_state[0]) * 1000003 + int(rng_state[1]) + 17)
        m = mu[:n, :act_dim].float()
        if mu_tanh:
            m = torch.tanh(m)
        s = torch.exp(logstd[:act_dim]).expand_as(m)
        a = m + s * torch.randn(n, act_dim, generator=g)
        nlp = 0.5 * (((a - m) / s) ** 2).sum(-1) + 0.5 * math.log(2 * math.pi) * act_dim + logstd[:act_dim].sum()
        keep = torch.ones(n)
        if rand_probs is not None:
            keep = torch.bernoulli(rand_probs[:n].float().cpu(), generator=g)
        mu_out[:n] = m
        sigma_out[:n] = s
        actions[:n] = torch.where(keep.view(-1, 1) != 0, a, m)
        neglogp.view(-1)[:n] = nlp
        if rand_mask is not None:
            rand_mask.view(-1)[:n] = keep
        rng_state[1] += 1

    def sample_latents(self, z, rows, dim, rng_state, row_offset=0, advance=True, z2=None):
        # counter-based like the kernel: the draw is a function of (seed, offset, global row), not of call history
        g = torch.Generator().manual_seed(int(rng_state[0]) * 1000003 + int(rng_state[1]))
        v = torch.randn(row_offset + rows, dim, generator=g)[row_offset:]
        z[:rows, :dim] = v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        if z2 is not None:
            z2[:rows, :dim] = z[:rows, :dim].to(z2.dtype)
        if advance:
            rng_state[1] += 1

    # ---- N2 (demo-side AMP observations): the oracle restatements stand in for the two kernels
    def motion_state(self, clips, motion_ids, times):
        from oracle import amp_obs as A
        c = dict(clips)
        for k in ('num_frames', 'length_starts'):
            c[k] = clips[k].long()
        return A.motion_state(c, motion_ids.long(), times)

    def build_amp_obs(self, root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_body_pos, dof_offsets,
                      local_root_obs, root_height_obs, hist, shift=True):
        from oracle import amp_obs as A
        frame = A.build_amp_observations(root_pos, root_rot, root_vel, root_ang_vel, dof_pos, dof_vel, key_body_pos,
                                         local_root_obs, root_height_obs, dof_offsets)
        if shift:
            A.push_history(hist, frame)
        else:
            hist[:, 0] = frame
