// Loss heads of the ASE/AMP/PPO update: forward value and analytic input-gradient in one pass,
// f32 math (as the reference), f64 accumulators for the reported scalars.
#include "common.h"
#include "act.h"

namespace {

constexpr float kHalfLog2Pi = 0.91893853320467274178f;  // 0.5 * ln(2*pi)

__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ x, int64_t n, int square,
                                                         double* __restrict__ acc) {
    __shared__ double sm[16];
    double v[1] = {0.0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double t = (double)x[i];
        v[0] += square ? t * t : t;
    }
    block_sum<1>(v, sm);
    if (threadIdx.x == 0) atomic_add_f64(acc, v[0]);
}

struct PpoArgs {
    const float* mu; int64_t ld_mu;
    const float* value; int64_t ld_v;
    const float *actions, *old_mu, *old_sigma, *old_logp, *adv, *old_value, *ret, *mask, *z, *new_z, *logstd;
    void* d_mu; int64_t ld_dmu;
    void* d_value; int64_t ld_dv;
    float *db_mu, *db_value, *mu_out;
    double* acc;
    double* scratch;              // [8 spare] then [gridDim.x][kPpoSlots] per-workgroup partial sums (no contended atomics)
    int M, m_global, act_dim, z_dim, masked, div_on, mu_tanh, clip_value;
    float e_clip, critic_coef, bounds_coef, div_coef, div_tar;
    float gs, inv_gs;             // the stored head gradients carry the gradient scale (f16 storage), the bias gradients do not
    float* gs_dev;                // nullable: scale record {factor on top of gs, overflow count} (the dynamic loss scale, common.h)
};

// per-workgroup partials: 7 loss sums, then 64 + 1 head-bias column sums (mu columns, value)
constexpr int kPpoSlots = 72;

// LPR lanes per row (act_dim <= LPR: 32 for the humanoid's 28 / 31 actions, 64 for the HRL high-level policy whose action
// is the 64-d latent), 256 / LPR rows per 256-thread block.
// (8 waves per SIMD = at most 64 registers, at the price of ~10 spilled dwords per lane: a latency-bound kernel on the step's critical
//  path that starts while other branches' matrix kernels hold every CU - their two waves per SIMD leave 64 registers (round 6), and at
//  72 this kernel waited for a CU to drain: 18 us alone, 111 us behind a resident grid)
template <typename T, int LPR>
__global__ __launch_bounds__(256, 8) void ppo_head_kernel(PpoArgs p) {
    constexpr int ROWS = 256 / LPR;
    __shared__ double sm[7 * 16];
    __shared__ float sdb[ROWS][LPR + 1];
    const int lane = threadIdx.x & (LPR - 1), rib = threadIdx.x / LPR;
    float gm_out = 0.f, gm2_out = 0.f, dv_out = 0.f;
    const int D = p.act_dim;
    if (p.gs_dev) {
        p.gs *= *p.gs_dev;
        p.inv_gs = 1.f / p.gs;
    }
    double part[7] = {0, 0, 0, 0, 0, 0, 0};  // a_loss, b_loss, entropy, clipped, c_loss, kl, div
    bool bad = false;                        // a stored head gradient overflowed

    // grid-stride over rows: few workgroups => few contended f64 atomics on the 7 accumulators
    for (int i = blockIdx.x * ROWS + rib; i < p.M; i += gridDim.x * ROWS) {
        const bool act_ok = lane < p.act_dim;
        const float S = p.masked ? (float)p.acc[ASE_ACC_MASK_SUM] : (float)p.m_global;
        const float mk = p.masked ? p.mask[i] : 1.f;
        const float w = mk / S;

        float m = 0.f, a = 0.f, ls = 0.f, sg = 1.f, d = 0.f, omu = 0.f, osg = 1.f;
        if (act_ok) {
            const float raw = p.mu[(int64_t)i * p.ld_mu + lane];
            m = p.mu_tanh ? tanhf(raw) : raw;
            a = p.actions[(int64_t)i * D + lane];
            ls = p.logstd[lane];
            sg = expf(ls);
            d = (a - m) / sg;
            omu = p.old_mu[(int64_t)i * D + lane];
            osg = p.old_sigma[(int64_t)i * D + lane];
            if (p.mu_out) p.mu_out[(int64_t)i * D + lane] = m;
        }
        const float sum_d2 = group_sum<LPR>(act_ok ? d * d : 0.f);
        const float sum_ls = group_sum<LPR>(act_ok ? ls : 0.f);
        const float nlp = 0.5f * sum_d2 + kHalfLog2Pi * (float)D + sum_ls;
        const float ratio = expf(p.old_logp[i] - nlp);
        const float adv = p.adv[i];
        const float rc = fminf(fmaxf(ratio, 1.f - p.e_clip), 1.f + p.e_clip);
        const float s1 = -adv * ratio, s2 = -adv * rc;
        const float a_loss = fmaxf(s1, s2);
        float g;  // d a_loss / d ratio  (torch.max splits ties evenly; clamp passes gradient inside [lo, hi])
        if (ratio == rc) g = -adv;
        else g = (s1 > s2) ? -adv : ((s1 == s2) ? -0.5f * adv : 0.f);

        // bound loss, entropy, KL (rl_games policy_kl(p0 = new, p1 = old))
        const float bh = fmaxf(m - 1.f, 0.f), bl = fminf(m + 1.f, 0.f);
        const float b_row = group_sum<LPR>(act_ok ? bh * bh + bl * bl : 0.f);
        const float ent_row = group_sum<LPR>(act_ok ? 0.5f + kHalfLog2Pi + ls : 0.f);
        float klj = 0.f;
        if (act_ok) {
            const float c1 = logf(osg / sg + 1e-5f);
            const float c2 = (sg * sg + (omu - m) * (omu - m)) / (2.f * (osg * osg + 1e-5f));
            klj = c1 + c2 - 0.5f;
        }
        const float kl_row = group_sum<LPR>(klj);

        float gm = w * g * ratio * d / sg + p.bounds_coef * w * 2.f * (bh + bl);

        // diversity (learning/ase_agent.py:445-467)
        float div_row = 0.f, gm2 = 0.f, m2 = 0.f;
        if (p.div_on) {
            float cm = 0.f, cm2 = 0.f;
            if (act_ok) {
                const float raw2 = p.mu[(int64_t)(p.M + i) * p.ld_mu + lane];
                m2 = p.mu_tanh ? tanhf(raw2) : raw2;
                cm = fminf(fmaxf(m, -1.f), 1.f);
                cm2 = fminf(fmaxf(m2, -1.f), 1.f);
            }
            const float diff = cm - cm2;
            const float a_diff = group_sum<LPR>(act_ok ? diff * diff : 0.f) / (float)D;
            float zz = 0.f;
            for (int k = lane; k < p.z_dim; k += LPR) zz += p.new_z[(int64_t)i * p.z_dim + k] * p.z[(int64_t)i * p.z_dim + k];
            zz = group_sum<LPR>(zz);
            const float z_diff = 0.5f - 0.5f * zz;
            const float inv = 1.f / (z_diff + 1e-5f);
            const float bonus = a_diff * inv;
            const float t = p.div_tar - bonus;
            div_row = t * t;
            const float dl_dbonus = p.div_coef * w * 2.f * (bonus - p.div_tar);
            const float db = dl_dbonus * 2.f * diff / (float)D * inv;
            if (act_ok) {
                if (m >= -1.f && m <= 1.f) gm += db;
                if (m2 >= -1.f && m2 <= 1.f) gm2 = -db;
            }
        }
        if (act_ok) {
            if (p.mu_tanh) {
                gm *= (1.f - m * m);
                gm2 *= (1.f - m2 * m2);
            }
            const T o1 = from_f32<T>(p.gs * gm), o2 = from_f32<T>(p.gs * gm2);
            reinterpret_cast<T*>(p.d_mu)[(int64_t)i * p.ld_dmu + lane] = o1;
            if (p.div_on) reinterpret_cast<T*>(p.d_mu)[(int64_t)(p.M + i) * p.ld_dmu + lane] = o2;
            bad |= ovf_hit1(o1) || (p.div_on && ovf_hit1(o2));
            gm_out += p.inv_gs * to_f32(o1);
            gm2_out += p.div_on ? p.inv_gs * to_f32(o2) : 0.f;
        }

        if (lane == 0) {
            // critic (learning/common_agent.py:521-534)
            const float v = p.value[(int64_t)i * p.ld_v];
            const float R = p.ret[i];
            float c, dv;
            if (p.clip_value) {
                const float ov = p.old_value[i];
                const float dlt = v - ov;
                const float vpc = ov + fminf(fmaxf(dlt, -p.e_clip), p.e_clip);
                const float l1 = (v - R) * (v - R), l2 = (vpc - R) * (vpc - R);
                const float inside = (dlt >= -p.e_clip && dlt <= p.e_clip) ? 1.f : 0.f;
                c = fmaxf(l1, l2);
                const float g1 = 2.f * (v - R), g2 = 2.f * (vpc - R) * inside;
                dv = (l1 > l2) ? g1 : ((l1 < l2) ? g2 : 0.5f * (g1 + g2));
            } else {
                c = (R - v) * (R - v);
                dv = 2.f * (v - R);
            }
            const T ov = from_f32<T>(p.gs * (p.critic_coef * dv / (float)p.m_global));
            reinterpret_cast<T*>(p.d_value)[(int64_t)i * p.ld_dv] = ov;
            bad |= ovf_hit1(ov);
            dv_out += p.inv_gs * to_f32(ov);
            part[0] += (double)(mk * a_loss);
            part[1] += (double)(mk * b_row);
            part[2] += (double)(mk * ent_row);
            part[3] += (double)(mk * ((fabsf(ratio - 1.f) > p.e_clip) ? 1.f : 0.f));
            part[4] += (double)c;
            part[5] += (double)kl_row;
            part[6] += (double)(mk * div_row);
        }
    }
    ovf_report(p.gs_dev, bad);
    // Every workgroup leaves its partial sums (7 loss sums in f64, the head-bias column sums of its rows) in its scratch
    // slab - 1024 workgroups adding to the same 40 addresses with atomics cost ~20 us of this kernel; ppo_head_fold_kernel
    // folds the slabs.
    double* const slabs = p.scratch + 8;
    double* mine = slabs + (int64_t)blockIdx.x * kPpoSlots;
    sdb[rib][lane] = gm_out + gm2_out;
    if (lane == 0) sdb[rib][LPR] = dv_out;
    __syncthreads();
    if (threadIdx.x < LPR + 1) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < ROWS; ++q) t += sdb[q][threadIdx.x];
        mine[7 + (threadIdx.x < LPR ? threadIdx.x : 64)] = (double)t;
    }
    block_sum<7>(part, sm);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) mine[k] = part[k];
    }
}

// second stage (a kernel boundary is the cheapest agent-scope release / acquire there is: a per-workgroup release fence
// - one L2 write-back each - made the 1024-workgroup kernel 6x slower): a few workgroups fold the slabs into the accumulators
// and the head-bias gradients.  Thread (c, g) sums slot c over the slabs b = lo + g, lo + g + 4, ...; the four groups meet in LDS.
__global__ __launch_bounds__(256) void ppo_head_fold_kernel(const double* __restrict__ slabs, int nblocks, double* __restrict__ acc,
                                                            float* __restrict__ db_mu, float* __restrict__ db_value, int act_dim,
                                                            int div_on) {
    // gridDim.x workgroups share the slabs; 8 independent loads in flight per thread (a single chain of dependent loads over
    // 1024 slabs took 129 us)
    __shared__ double fold[4][kPpoSlots];
    const int c = threadIdx.x & 63, g4 = threadIdx.x >> 6;
    const int per = (nblocks + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = min(nblocks, lo + per);
    for (int cc = c; cc < kPpoSlots; cc += 64) {
        double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int b = lo + g4;
        for (; b + 28 < hi; b += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] += slabs[(int64_t)(b + 4 * u) * kPpoSlots + cc];
        }
        for (; b < hi; b += 4) t[0] += slabs[(int64_t)b * kPpoSlots + cc];
        fold[g4][cc] = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    __syncthreads();
    if (threadIdx.x < kPpoSlots) {
        const int cc = threadIdx.x;
        const double t = fold[0][cc] + fold[1][cc] + fold[2][cc] + fold[3][cc];
        if (cc < 7) {
            const int slot[7] = {ASE_ACC_A_LOSS, ASE_ACC_B_LOSS, ASE_ACC_ENTROPY, ASE_ACC_CLIPPED, ASE_ACC_C_LOSS, ASE_ACC_KL, ASE_ACC_DIV};
            if (cc < 6 || div_on) atomic_add_f64(acc + slot[cc], t);
        } else if (db_mu) {
            const int j = cc - 7;
            if (j < act_dim) atomic_add_f32(db_mu + j, (float)t);
            else if (j == 64 && db_value) atomic_add_f32(db_value, (float)t);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void disc_head_kernel(const float* __restrict__ logit, int64_t ld_l, T* __restrict__ d_logit,
                                                        int64_t ld_d, float* __restrict__ db_logit, double* __restrict__ acc,
                                                        int amb, int amb_global, float disc_coef, float gs,
                                                        float* __restrict__ gs_dev) {
    __shared__ double sm[5 * 16];
    if (gs_dev) gs *= *gs_dev;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    double part[5] = {0, 0, 0, 0, 0};  // bce agent, bce demo, agent acc, demo acc, sum of d_logit
    bool bad = false;
    if (r < 3 * amb) {
        const float l = logit[(int64_t)r * ld_l];
        const float sp_abs = log1pf(expf(-fabsf(l)));
        float g;
        if (r < 2 * amb) {  // BCE(l, 0) = softplus(l)
            part[0] = (double)(fmaxf(l, 0.f) + sp_abs);
            part[2] = (l < 0.f) ? 1.0 : 0.0;
            const float sig = 1.f / (1.f + expf(-l));
            g = disc_coef * 0.5f * sig / (2.f * (float)amb_global);
        } else {            // BCE(l, 1) = softplus(-l)
            part[1] = (double)(fmaxf(-l, 0.f) + sp_abs);
            part[3] = (l > 0.f) ? 1.0 : 0.0;
            const float sig_neg = 1.f / (1.f + expf(l));
            g = -disc_coef * 0.5f * sig_neg / (float)amb_global;
        }
        const T o = from_f32<T>(gs * g);
        d_logit[(int64_t)r * ld_d] = o;
        bad = ovf_hit1(o);
        part[4] = (double)(to_f32(o) / gs);
    }
    ovf_report(gs_dev, bad);
    block_sum<5>(part, sm);
    if (threadIdx.x == 0) {
        if (db_logit) atomic_add_f32(db_logit, (float)part[4]);
        atomic_add_f64(acc + ASE_ACC_BCE_AGENT, part[0]);
        atomic_add_f64(acc + ASE_ACC_BCE_DEMO, part[1]);
        atomic_add_f64(acc + ASE_ACC_AGENT_ACC, part[2]);
        atomic_add_f64(acc + ASE_ACC_DEMO_ACC, part[3]);
    }
}

// one wave per row; z_dim <= 128
template <typename T>
__global__ __launch_bounds__(256) void enc_head_kernel(const float* __restrict__ e, int64_t ld_e, const float* __restrict__ z,
                                                       int64_t ld_z, T* __restrict__ d_e, int64_t ld_de,
                                                       float* __restrict__ db_enc, float* __restrict__ enc_out,
                                                       double* __restrict__ acc, int amb, int amb_global, int z_dim,
                                                       float enc_coef, float gs, float* __restrict__ gs_dev) {
    __shared__ double sm[16];
    if (gs_dev) gs *= *gs_dev;
    __shared__ float sdb[4][128];
    float dbv[2] = {0.f, 0.f};
    const int lane = threadIdx.x & 63;
    double part[1] = {0.0};
    bool bad = false;
    // grid-stride over rows: <= 128 workgroups, so the 64 bias-gradient addresses and the loss accumulator see <= 128
    // atomics each (one workgroup per 4 rows = 1024 contended atomics per address: 22 us for 4096 rows)
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < amb; r += gridDim.x * 4) {
        float ev[2] = {0.f, 0.f}, zv[2] = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = lane + 64 * q;
            if (j < z_dim) {
                ev[q] = e[(int64_t)r * ld_e + j];
                zv[q] = z[(int64_t)r * ld_z + j];
            }
        }
        const float ss = wave_sum(ev[0] * ev[0] + ev[1] * ev[1]);
        const float nrm = fmaxf(sqrtf(ss), 1e-12f);
        const float h0 = ev[0] / nrm, h1 = ev[1] / nrm;
        const float dot = wave_sum(h0 * zv[0] + h1 * zv[1]);
        const float sc = enc_coef / (float)amb_global;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = lane + 64 * q;
            if (j < z_dim) {
                const float h = q ? h1 : h0;
                const T o = from_f32<T>(gs * (-sc * (zv[q] - h * dot) / nrm));
                d_e[(int64_t)r * ld_de + j] = o;
                bad |= ovf_hit1(o);
                dbv[q] += to_f32(o) / gs;
                if (enc_out) enc_out[(int64_t)r * z_dim + j] = h;
            }
        }
        if (lane == 0) part[0] += (double)(-dot);
    }
    ovf_report(gs_dev, bad);
    if (db_enc) {
        sdb[threadIdx.x >> 6][lane] = dbv[0];
        sdb[threadIdx.x >> 6][lane + 64] = dbv[1];
        __syncthreads();
        if (threadIdx.x < z_dim) {
            const float t = sdb[0][threadIdx.x] + sdb[1][threadIdx.x] + sdb[2][threadIdx.x] + sdb[3][threadIdx.x];
            atomic_add_f32(db_enc + threadIdx.x, t);
        }
    }
    block_sum<1>(part, sm);
    if (threadIdx.x == 0) atomic_add_f64(acc + ASE_ACC_ENC, part[0]);
}

// Encoder gradient penalty (learning/ase_agent.py:431-441), seed and return of the chain.  Per row, with n = |e|,
// eh = e / n, a = eh . z, the error is err = -a and
//   u = d err / d e = -(z - eh a) / n                                   (seed of the chain W_e^T u -> ... -> d err / d x)
//   J = d u / d e  (symmetric):  J r = [z (eh . r) + eh (z . r) + a r - 3 a eh (eh . r)] / n^2
// MODE 0: u_out = scale * u.   MODE 1: d_e += J du (du = what the chain's backward returns at u), the bias gradient
// receives the change of the STORED d_e (db == column sums of what is stored).  One wave per row, z_dim <= 128.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void enc_gp_kernel(const float* __restrict__ e, int64_t ld_e, const float* __restrict__ z,
                                                     int64_t ld_z, const float* __restrict__ du, int64_t ld_du,
                                                     T* __restrict__ out, int64_t ld_out, float* __restrict__ db_enc,
                                                     int rows, int z_dim, float scale, float* __restrict__ scale_dev) {
    __shared__ float sdb[4][128];
    if (scale_dev) scale *= *scale_dev;
    float dbv[2] = {0.f, 0.f};
    bool bad = false;
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += gridDim.x * 4) {
        float ev[2] = {0.f, 0.f}, zv[2] = {0.f, 0.f}, dv[2] = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = lane + 64 * q;
            if (j < z_dim) {
                ev[q] = e[(int64_t)r * ld_e + j];
                zv[q] = z[(int64_t)r * ld_z + j];
                if (MODE == 1) dv[q] = du[(int64_t)r * ld_du + j];
            }
        }
        const float nrm = fmaxf(sqrtf(wave_sum(ev[0] * ev[0] + ev[1] * ev[1])), 1e-12f);
        const float h0 = ev[0] / nrm, h1 = ev[1] / nrm;
        const float a = wave_sum(h0 * zv[0] + h1 * zv[1]);
        float hr = 0.f, zr = 0.f;
        if (MODE == 1) {
            hr = wave_sum(h0 * dv[0] + h1 * dv[1]);
            zr = wave_sum(zv[0] * dv[0] + zv[1] * dv[1]);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = lane + 64 * q;
            if (j < z_dim) {
                const float h = q ? h1 : h0;
                if (MODE == 0) {
                    const T o = from_f32<T>(-scale * (zv[q] - h * a) / nrm);
                    out[(int64_t)r * ld_out + j] = o;
                    bad |= ovf_hit1(o);
                } else {
                    const float jr = (zv[q] * hr + h * zr + a * dv[q] - 3.f * a * h * hr) / (nrm * nrm);
                    const float old = to_f32(out[(int64_t)r * ld_out + j]);      // (carries the gradient scale)
                    const T nw = from_f32<T>(old + scale * jr);
                    out[(int64_t)r * ld_out + j] = nw;
                    bad |= ovf_hit1(nw);
                    dbv[q] += (to_f32(nw) - old) / scale;
                }
            }
        }
    }
    ovf_report(scale_dev, bad);
    if (MODE == 1 && db_enc) {
        sdb[threadIdx.x >> 6][lane] = dbv[0];
        sdb[threadIdx.x >> 6][lane + 64] = dbv[1];
        __syncthreads();
        if (threadIdx.x < z_dim) {
            const float t = sdb[0][threadIdx.x] + sdb[1][threadIdx.x] + sdb[2][threadIdx.x] + sdb[3][threadIdx.x];
            atomic_add_f32(db_enc + threadIdx.x, t);
        }
    }
}

// derivative factors of the gradient-penalty chain from a layer's twin t: the activation OUTPUT for ReLU / tanh (both
// derivatives are functions of it), the PRE-activation for the smooth activations
__device__ __forceinline__ float twin_grad(int act, float t) {
    if (act == ASE_ACT_RELU) return t > 0.f ? 1.f : 0.f;
    if (act == ASE_ACT_TANH) return 1.f - t * t;
    return act_grad(act, t);
}
// second-order factor act''(z) u r of the gradient penalty from the stored chain values g = act' u and dg = act' r:
// (act'' / act') * (g / act') * dg, two separate divisions - act'^2 underflows to 0 for saturated sigmoid / ELU / SELU /
// GELU units (|z| of a few tens) while act' itself is still a normal number, and act'' / act'^2 was inf there.  0 where act'
// vanishes (the chain value it multiplies is 0 there too), and 0 where FINITE chain values overflow through a tiny act' (the
// underflow case above); a NaN / inf that ARRIVES in g or dg - a diverging penalty chain - is passed on, not swallowed.
__device__ __forceinline__ float twin_second(int act, float t, float g, float dg) {
    float d1, d2;
    if (act == ASE_ACT_RELU || act == ASE_ACT_NONE) return 0.f;
    if (act == ASE_ACT_TANH) { d1 = 1.f - t * t; d2 = -2.f * t * d1; }
    else { d1 = act_grad(act, t); d2 = act_grad2(act, t); }
    if (d1 == 0.f) return 0.f;
    const float e = (d2 / d1) * (g / d1) * dg;
    const bool in_finite = fabsf(g) <= 3.0e38f && fabsf(dg) <= 3.0e38f;          // (false for NaN)
    return (!in_finite || (e == e && fabsf(e) <= 3.0e38f)) ? e : 0.f;
}

template <typename T>
__global__ __launch_bounds__(256) void gp_seed_kernel(const T* __restrict__ h, int64_t ld_h, const float* __restrict__ w,
                                                      T* __restrict__ g, int64_t ld_g, int rows, int width, float scale, int act) {
    const int64_t n = (int64_t)rows * width;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / width), j = (int)(i - (int64_t)r * width);
        const float hv = to_f32(h[(int64_t)r * ld_h + j]);
        g[(int64_t)r * ld_g + j] = from_f32<T>(scale * w[j] * twin_grad(act, hv));
    }
}

// second-order term of the gradient penalty's backward: dz[r, j] += act''(z) / act'(z)^2 * g[r, j] * dg[r, j]
// (g = act' u the chain value, dg = act' r the masked backward-of-chain value: the product is act'' u r)
template <typename T>
__global__ __launch_bounds__(256) void gp_second_kernel(const T* __restrict__ t, int64_t ld_t, const T* __restrict__ g, int64_t ld_g,
                                                        const T* __restrict__ dg, int64_t ld_dg, T* __restrict__ dz, int64_t ld_dz,
                                                        int rows, int width, int act) {
    const int64_t n = (int64_t)rows * width;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / width), j = (int)(i - (int64_t)r * width);
        const float e = twin_second(act, to_f32(t[(int64_t)r * ld_t + j]), to_f32(g[(int64_t)r * ld_g + j]),
                                    to_f32(dg[(int64_t)r * ld_dg + j]));
        T* o = dz + (int64_t)r * ld_dz + j;
        *o = from_f32<T>(to_f32(*o) + e);
    }
}

// 16-byte row chunks (8 bf16 / 4 f32), f32 partial per chunk, f64 across chunks; cols % VEC == 0 and 16-byte aligned rows
// (the host falls back to VEC = 1 otherwise)
template <typename T, int VEC>
__global__ __launch_bounds__(256) void sqnorm_kernel(const T* __restrict__ x, int64_t ld, int rows, int cols,
                                                     double* __restrict__ acc, double scale, const float* __restrict__ scale_dev) {
    __shared__ double sm[16];
    if (scale_dev) scale *= (double)*scale_dev;
    const int cpr = cols / VEC;                            // chunks per row
    const int64_t n = (int64_t)rows * cpr;
    double v[1] = {0.0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / cpr), j = (int)(i - (int64_t)r * cpr) * VEC;
        const T* p = x + (int64_t)r * ld + j;
        float s = 0.f;
        if constexpr (VEC == 8) {
            const typename V16<T>::x8 q = *reinterpret_cast<const typename V16<T>::x8*>(p);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float t = (float)q[e]; s += t * t; }
        } else if constexpr (VEC == 4) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(p);
#pragma unroll
            for (int e = 0; e < 4; ++e) s += q[e] * q[e];
        } else {
            const float t = to_f32(p[0]);
            s = t * t;
        }
        v[0] += (double)s;
    }
    block_sum<1>(v, sm);
    if (threadIdx.x == 0) atomic_add_f64(acc, v[0] * scale);
}

// out[j] += scale * sum_r x[r, j] (f32 matrix, row pitch ld): 64 columns x 4 row groups per workgroup, grid-stride over
// row chunks, one atomic per column and workgroup
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ld, int rows, int cols, float scale,
                                                     float* __restrict__ out, int rows_per_block) {
    __shared__ float red[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tx;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (j < cols) {
        int r = r0 + ty;
        for (; r + 12 < r1; r += 16) {
            s0 += x[(int64_t)r * ld + j]; s1 += x[(int64_t)(r + 4) * ld + j];
            s2 += x[(int64_t)(r + 8) * ld + j]; s3 += x[(int64_t)(r + 12) * ld + j];
        }
        for (; r < r1; r += 4) s0 += x[(int64_t)r * ld + j];
    }
    red[ty][tx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ty == 0 && j < cols) atomic_add_f32(out + j, scale * (red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx]));
}

struct FinArgs {
    int m_global, amb_global, masked, has_disc, has_enc, has_div;
    double* opt_state;        // nullable: adaptive learning rate (rl_games AdaptiveScheduler) - opt_state[1] is the lr
    float kl_threshold;
    float critic_coef, entropy_coef, bounds_coef, disc_coef, disc_logit_reg, disc_grad_penalty, disc_weight_decay,
        enc_coef, enc_weight_decay, div_coef, enc_grad_penalty;
};

__global__ void finalize_scalars_kernel(const double* __restrict__ acc, float* __restrict__ out, FinArgs a) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double S = acc[ASE_ACC_MASK_SUM];
    const double den = a.masked ? S : (double)a.m_global;
    const double al = acc[ASE_ACC_A_LOSS] / den, bl = acc[ASE_ACC_B_LOSS] / den, ent = acc[ASE_ACC_ENTROPY] / den;
    const double cf = acc[ASE_ACC_CLIPPED] / den;
    const double cl = acc[ASE_ACC_C_LOSS] / (double)a.m_global, kl = acc[ASE_ACC_KL] / (double)a.m_global;
    double loss = al + a.critic_coef * cl - a.entropy_coef * ent + a.bounds_coef * bl;
    for (int i = 0; i < ASE_RES_COUNT; ++i) out[i] = 0.f;
    out[ASE_RES_A_LOSS] = (float)al;
    out[ASE_RES_C_LOSS] = (float)cl;
    out[ASE_RES_B_LOSS] = (float)bl;
    out[ASE_RES_ENTROPY] = (float)ent;
    out[ASE_RES_CLIP_FRAC] = (float)cf;
    out[ASE_RES_KL] = (float)kl;
    out[ASE_RES_MASK_SUM] = (float)S;
    if (a.has_disc) {
        const double amb = (double)a.amb_global;
        const double bce = 0.5 * (acc[ASE_ACC_BCE_AGENT] / (2.0 * amb) + acc[ASE_ACC_BCE_DEMO] / amb);
        const double gp = acc[ASE_ACC_GP] / amb;
        const double dl = bce + a.disc_logit_reg * acc[ASE_ACC_LOGIT_W2] + a.disc_grad_penalty * gp +
                          a.disc_weight_decay * acc[ASE_ACC_DISC_W2];
        loss += a.disc_coef * dl;
        out[ASE_RES_DISC_LOSS] = (float)dl;
        out[ASE_RES_DISC_GP] = (float)gp;
        out[ASE_RES_DISC_LOGIT_LOSS] = (float)acc[ASE_ACC_LOGIT_W2];
        out[ASE_RES_DISC_AGENT_ACC] = (float)(acc[ASE_ACC_AGENT_ACC] / (2.0 * amb));
        out[ASE_RES_DISC_DEMO_ACC] = (float)(acc[ASE_ACC_DEMO_ACC] / amb);
    }
    if (a.has_enc) {
        const double egp = acc[ASE_ACC_ENC_GP] / (double)a.amb_global;
        const double el = acc[ASE_ACC_ENC] / (double)a.amb_global + a.enc_weight_decay * acc[ASE_ACC_ENC_W2] +
                          a.enc_grad_penalty * egp;
        loss += a.enc_coef * el;
        out[ASE_RES_ENC_LOSS] = (float)el;
        out[ASE_RES_ENC_GP] = (float)egp;
    }
    if (a.has_div) {
        const double dv = acc[ASE_ACC_DIV] / S;
        loss += a.div_coef * dv;
        out[ASE_RES_DIV_LOSS] = (float)dv;
    }
    out[ASE_RES_LOSS] = (float)loss;
    if (a.opt_state) {
        // rl_games schedulers.AdaptiveScheduler.update (min_lr 1e-6, max_lr 1e-2), 'legacy' schedule: after every optimisation
        // step, from that step's kl (learning/common_agent.py:204-208)
        double lr = a.opt_state[1];
        const double cur = lr;
        out[ASE_RES_LR] = (float)cur;          // the rate this step was taken with (train_result['last_lr'])
        if (kl > 2.0 * (double)a.kl_threshold) lr = fmax(cur / 1.5, 1e-6);
        if (kl < 0.5 * (double)a.kl_threshold) lr = fmin(cur * 1.5, 1e-2);
        a.opt_state[1] = lr;
    }
}

inline int grid_for(int64_t n, int block = 256, int cap = 2048) {
    int64_t g = (n + block - 1) / block;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

}  // namespace

extern "C" int ase_hip_reduce_sum(const float* x, int64_t n, int square, double* acc, int slot, void* stream) {
    ASE_CHECK_ARG(x && acc && n > 0 && slot >= 0, "reduce_sum: null/empty operand");
    ASE_LAUNCH(reduce_sum_kernel, dim3(grid_for(n, 1024, 128)), dim3(256), 0, (hipStream_t)stream, x, n, square,
                       acc + slot);
    ASE_CHECK_LAUNCH("reduce_sum");
    return ASE_OK;
}

extern "C" int ase_hip_ppo_head(const float* mu, int64_t ld_mu, const float* value, int64_t ld_v,
                                const float* mb_actions, const float* mb_old_mu, const float* mb_old_sigma,
                                const float* mb_old_logp, const float* mb_adv, const float* mb_old_value,
                                const float* mb_return, const float* mb_mask, const float* mb_z, const float* new_z,
                                const float* logstd, void* d_mu, int64_t ld_dmu, void* d_value, int64_t ld_dv,
                                float* db_mu, float* db_value, float* mu_out, double* acc, double* scratch, int M, int m_global, int act_dim, int z_dim, int masked,
                                int div_on, int mu_tanh, int clip_value, float e_clip, float critic_coef,
                                float bounds_coef, float div_coef, float div_tar, float grad_scale, float* grad_scale_dev, int dtype,
                                void* stream) {
    ASE_CHECK_ARG(mu && value && mb_actions && mb_old_mu && mb_old_sigma && mb_old_logp && mb_adv && mb_return &&
                      logstd && d_mu && d_value && acc && scratch && M > 0 && m_global >= M,
                  "ppo_head: null/empty operand");
    ASE_CHECK_ARG(act_dim >= 1 && act_dim <= 64, "ppo_head: act_dim %d not in [1,64]", act_dim);
    ASE_CHECK_ARG(!masked || mb_mask, "ppo_head: masked reduction without a mask");
    ASE_CHECK_ARG(!div_on || (mb_z && new_z && z_dim > 0), "ppo_head: diversity loss without latents");
    ASE_CHECK_ARG(!clip_value || mb_old_value, "ppo_head: clip_value without old values");
    ASE_CHECK_ARG(grad_scale > 0.f, "ppo_head: grad_scale must be positive");
    PpoArgs p;
    p.mu = mu; p.ld_mu = ld_mu; p.value = value; p.ld_v = ld_v;
    p.actions = mb_actions; p.old_mu = mb_old_mu; p.old_sigma = mb_old_sigma; p.old_logp = mb_old_logp;
    p.adv = mb_adv; p.old_value = mb_old_value; p.ret = mb_return; p.mask = mb_mask; p.z = mb_z; p.new_z = new_z;
    p.logstd = logstd; p.d_mu = d_mu; p.ld_dmu = ld_dmu; p.d_value = d_value; p.ld_dv = ld_dv; p.db_mu = db_mu; p.db_value = db_value; p.mu_out = mu_out;
    p.acc = acc; p.scratch = scratch; p.M = M; p.m_global = m_global; p.act_dim = act_dim; p.z_dim = z_dim; p.masked = masked;
    p.div_on = div_on; p.mu_tanh = mu_tanh; p.clip_value = clip_value; p.e_clip = e_clip; p.critic_coef = critic_coef;
    p.bounds_coef = bounds_coef; p.div_coef = div_coef; p.div_tar = div_tar; p.gs = grad_scale; p.inv_gs = 1.f / grad_scale; p.gs_dev = grad_scale_dev;
    const int rows = act_dim <= 32 ? 8 : 4;         // rows per workgroup (32 / 64 lanes per row)
    const dim3 grid(min((M + rows - 1) / rows, 1024));       // scratch: (1024 x 72 + 1) doubles, the ticket word zero between launches
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        if (act_dim <= 32) ASE_LAUNCH((ppo_head_kernel<T, 32>), grid, dim3(256), 0, (hipStream_t)stream, p);
        else ASE_LAUNCH((ppo_head_kernel<T, 64>), grid, dim3(256), 0, (hipStream_t)stream, p);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "ppo_head: bad dtype %d", dtype);
    ASE_LAUNCH(ppo_head_fold_kernel, dim3(grid.x >= 64 ? 8 : 1), dim3(256), 0, (hipStream_t)stream, (const double*)(scratch + 8), (int)grid.x, acc, db_mu,
               db_value, act_dim, div_on);
    ASE_CHECK_LAUNCH("ppo_head");
    return ASE_OK;
}

extern "C" int ase_hip_disc_head(const float* logit, int64_t ld_l, void* d_logit, int64_t ld_d, float* db_logit,
                                 double* acc, int amb, int amb_global, float disc_coef, float grad_scale,
                                 float* grad_scale_dev, int dtype, void* stream) {
    ASE_CHECK_ARG(logit && d_logit && acc && amb > 0 && amb_global >= amb && grad_scale > 0.f, "disc_head: null/empty operand");
    const dim3 grid((3 * amb + 255) / 256);
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH(disc_head_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, logit, ld_l, (T*)d_logit, ld_d, db_logit, acc, amb,
                   amb_global, disc_coef, grad_scale, grad_scale_dev);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "disc_head: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("disc_head");
    return ASE_OK;
}

extern "C" int ase_hip_enc_head(const float* e, int64_t ld_e, const float* z, int64_t ld_z, void* d_e, int64_t ld_de,
                                float* db_enc, float* enc_out, double* acc, int amb, int amb_global, int z_dim, float enc_coef,
                                float grad_scale, float* grad_scale_dev, int dtype, void* stream) {
    ASE_CHECK_ARG(e && z && d_e && acc && amb > 0 && amb_global >= amb && grad_scale > 0.f, "enc_head: null/empty operand");
    ASE_CHECK_ARG(z_dim >= 1 && z_dim <= 128, "enc_head: z_dim %d not in [1,128]", z_dim);
    const dim3 grid(min((amb + 3) / 4, 128));
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH(enc_head_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, e, ld_e, z, ld_z, (T*)d_e, ld_de, db_enc, enc_out, acc,
                   amb, amb_global, z_dim, enc_coef, grad_scale, grad_scale_dev);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "enc_head: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("enc_head");
    return ASE_OK;
}

extern "C" int ase_hip_enc_gp_seed(const float* e, int64_t ld_e, const float* z, int64_t ld_z, void* u, int64_t ld_u, int rows,
                                   int z_dim, float scale, int dtype, void* stream) {
    ASE_CHECK_ARG(e && z && u && rows > 0, "enc_gp_seed: null/empty operand");
    ASE_CHECK_ARG(z_dim >= 1 && z_dim <= 128, "enc_gp_seed: z_dim %d not in [1,128]", z_dim);
    const dim3 grid(min((rows + 3) / 4, 1024));
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH((enc_gp_kernel<T, 0>), grid, dim3(256), 0, (hipStream_t)stream, e, ld_e, z, ld_z, (const float*)nullptr,
                   (int64_t)0, (T*)u, ld_u, (float*)nullptr, rows, z_dim, scale, (float*)nullptr);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "enc_gp_seed: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("enc_gp_seed");
    return ASE_OK;
}

extern "C" int ase_hip_enc_gp_back(const float* e, int64_t ld_e, const float* z, int64_t ld_z, const float* du, int64_t ld_du,
                                   void* d_e, int64_t ld_de, float* db_enc, int rows, int z_dim, float grad_scale,
                                   float* grad_scale_dev, int dtype, void* stream) {
    ASE_CHECK_ARG(e && z && du && d_e && rows > 0 && grad_scale > 0.f, "enc_gp_back: null/empty operand");
    ASE_CHECK_ARG(z_dim >= 1 && z_dim <= 128, "enc_gp_back: z_dim %d not in [1,128]", z_dim);
    const dim3 grid(min((rows + 3) / 4, 128));          // <= 128 workgroups on the bias-gradient atomics (see enc_head)
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH((enc_gp_kernel<T, 1>), grid, dim3(256), 0, (hipStream_t)stream, e, ld_e, z, ld_z, du, ld_du, (T*)d_e, ld_de,
                   db_enc, rows, z_dim, grad_scale, grad_scale_dev);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "enc_gp_back: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("enc_gp_back");
    return ASE_OK;
}

extern "C" int ase_hip_gp_seed(const void* h, int64_t ld_h, const float* w, void* g, int64_t ld_g, int rows, int width,
                               float scale, int act, int dtype, void* stream) {
    ASE_CHECK_ARG(h && w && g && rows > 0 && width > 0, "gp_seed: null/empty operand");
    ASE_CHECK_ARG(act >= ASE_ACT_RELU && act <= ASE_ACT_SOFTPLUS, "gp_seed: activation %d", act);
    const dim3 grid(grid_for((int64_t)rows * width));
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH(gp_seed_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)h, ld_h, w, (T*)g, ld_g, rows, width, scale,
                   act);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "gp_seed: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("gp_seed");
    return ASE_OK;
}

extern "C" int ase_hip_gp_second(const void* twin, int64_t ld_t, const void* g, int64_t ld_g, const void* dg, int64_t ld_dg,
                                 void* dz, int64_t ld_dz, int rows, int width, int act, int dtype, void* stream) {
    ASE_CHECK_ARG(twin && g && dg && dz && rows > 0 && width > 0, "gp_second: null/empty operand");
    ASE_CHECK_ARG(act >= ASE_ACT_RELU && act <= ASE_ACT_SOFTPLUS, "gp_second: activation %d", act);
    const dim3 grid(grid_for((int64_t)rows * width));
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH(gp_second_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, (const T*)twin, ld_t, (const T*)g, ld_g, (const T*)dg,
                   ld_dg, (T*)dz, ld_dz, rows, width, act);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "gp_second: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("gp_second");
    return ASE_OK;
}

extern "C" int ase_hip_sqnorm(const void* x, int64_t ld, int rows, int cols, double* acc, int slot, double scale,
                              const float* scale_dev, int dtype, void* stream) {
    ASE_CHECK_ARG(x && acc && rows > 0 && cols > 0 && slot >= 0, "sqnorm: null/empty operand");
    const int es = ase_elem_size(dtype), vec = 16 / es;
    const bool wide = cols % vec == 0 && (ld * es) % 16 == 0 && ((uintptr_t)x % 16) == 0;
    const dim3 grid(grid_for((int64_t)rows * cols / (wide ? vec : 1), 2048, 256));
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        constexpr int VEC = 16 / (int)sizeof(T);
        if (wide) ASE_LAUNCH((sqnorm_kernel<T, VEC>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ld, rows, cols, acc + slot, scale, scale_dev);
        else ASE_LAUNCH((sqnorm_kernel<T, 1>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ld, rows, cols, acc + slot, scale, scale_dev);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "sqnorm: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("sqnorm");
    return ASE_OK;
}

extern "C" int ase_hip_colsum(const float* x, int64_t ld, int rows, int cols, float scale, float* out, void* stream) {
    ASE_CHECK_ARG(x && out && rows > 0 && cols > 0 && ld >= cols, "colsum: null/empty operand");
    const int rpb = 128;
    ASE_LAUNCH(colsum_kernel, dim3((cols + 63) / 64, (rows + rpb - 1) / rpb), dim3(256), 0, (hipStream_t)stream, x, ld, rows, cols, scale,
               out, rpb);
    ASE_CHECK_LAUNCH("colsum");
    return ASE_OK;
}

extern "C" int ase_hip_finalize_scalars(const double* acc, float* out, int m_global, int amb_global, int masked,
                                        int has_disc, int has_enc, int has_div, float critic_coef, float entropy_coef,
                                        float bounds_coef, float disc_coef, float disc_logit_reg,
                                        float disc_grad_penalty, float disc_weight_decay, float enc_coef,
                                        float enc_weight_decay, float div_coef, float enc_grad_penalty, double* opt_state_lr,
                                        float kl_threshold, void* stream) {
    ASE_CHECK_ARG(acc && out && m_global > 0, "finalize_scalars: null/empty operand");
    ASE_CHECK_ARG(opt_state_lr == nullptr || kl_threshold > 0.f, "finalize_scalars: adaptive lr needs a positive kl threshold");
    FinArgs a;
    a.m_global = m_global; a.amb_global = amb_global; a.masked = masked; a.has_disc = has_disc; a.has_enc = has_enc;
    a.has_div = has_div; a.critic_coef = critic_coef; a.entropy_coef = entropy_coef; a.bounds_coef = bounds_coef;
    a.disc_coef = disc_coef; a.disc_logit_reg = disc_logit_reg; a.disc_grad_penalty = disc_grad_penalty;
    a.disc_weight_decay = disc_weight_decay; a.enc_coef = enc_coef; a.enc_weight_decay = enc_weight_decay;
    a.div_coef = div_coef; a.enc_grad_penalty = enc_grad_penalty; a.opt_state = opt_state_lr; a.kl_threshold = kl_threshold;
    ASE_LAUNCH(finalize_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, out, a);
    ASE_CHECK_LAUNCH("finalize_scalars");
    return ASE_OK;
}
