// Once-per-epoch rollout tail: AMP rewards, GAE, advantage normalisation, replay ring, latent RNG.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void disc_reward_kernel(const float* __restrict__ logit, int64_t ld_l,
                                                          float* __restrict__ r, int64_t n, float scale) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float l = logit[i * ld_l];
        const float prob = 1.f / (1.f + expf(-l));
        r[i] = -logf(fmaxf(1.f - prob, 0.0001f)) * scale;
    }
}

// one wave per row, z_dim <= 128
__global__ __launch_bounds__(256) void enc_reward_kernel(const float* __restrict__ e, int64_t ld_e,
                                                         const float* __restrict__ z, int64_t ld_z,
                                                         float* __restrict__ r, int64_t n, int z_dim, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * (int64_t)4 + (threadIdx.x >> 6);
    if (row >= n) return;
    float ev[2] = {0.f, 0.f}, zv[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        if (j < z_dim) {
            ev[q] = e[row * ld_e + j];
            zv[q] = z[row * ld_z + j];
        }
    }
    const float nrm = fmaxf(sqrtf(wave_sum(ev[0] * ev[0] + ev[1] * ev[1])), 1e-12f);
    const float dot = wave_sum((ev[0] / nrm) * zv[0] + (ev[1] / nrm) * zv[1]);
    if (lane == 0) r[row] = fmaxf(dot, 0.f) * scale;
}

__global__ __launch_bounds__(256) void gae_kernel(const uint8_t* __restrict__ dones, const float* __restrict__ values,
                                                  const float* __restrict__ next_values, const float* __restrict__ r_task,
                                                  const float* __restrict__ r_disc, const float* __restrict__ r_enc,
                                                  float w_task, float w_disc, float w_enc, float gamma, float gamma_tau,
                                                  float* __restrict__ advs, float* __restrict__ returns, int H, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float last = 0.f;
    for (int t = H - 1; t >= 0; --t) {
        const int64_t i = (int64_t)t * N + n;
        float r = w_task * r_task[i];
        if (r_disc) r = r + w_disc * r_disc[i];
        if (r_enc) r = r + w_enc * r_enc[i];
        const float not_done = 1.f - (float)dones[i];
        const float v = values[i];
        const float delta = r + gamma * next_values[i] - v;
        last = delta + gamma_tau * not_done * last;
        advs[i] = last;
        returns[i] = last + v;
    }
}

__global__ __launch_bounds__(256) void adv_moments_kernel(const float* __restrict__ ret, const float* __restrict__ val,
                                                          const float* __restrict__ mask, double* __restrict__ acc3,
                                                          int64_t n) {
    __shared__ double sm[3 * 16];
    double v[3] = {0, 0, 0};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = ret[i] - val[i];
        const float m = mask ? mask[i] : 1.f;
        const double am = (double)(a * m);
        v[0] += (double)m;
        v[1] += am;
        v[2] += am * am;
    }
    block_sum<3>(v, sm);
    if (threadIdx.x == 0) {
        atomic_add_f64(acc3 + 0, v[0]);
        atomic_add_f64(acc3 + 1, v[1]);
        atomic_add_f64(acc3 + 2, v[2]);
    }
}

__global__ __launch_bounds__(256) void adv_apply_kernel(const float* __restrict__ ret, const float* __restrict__ val,
                                                        float* __restrict__ adv, const double* __restrict__ acc3,
                                                        int64_t n, int normalize, int has_mask) {
    float mean = 0.f, denom = 1.f;
    if (normalize) {
        const double S = acc3[0];
        const double mu = acc3[1] / S;
        // masked: rl_games get_mean_std_with_masks; unmasked: (x - mean) / (x.std() + 1e-8), both unbiased
        const double min_sqr = acc3[2] / S - mu * mu;
        mean = (float)mu;
        denom = (float)sqrt(min_sqr * S / (S - 1.0)) + 1e-8f;
        (void)has_mask;
    }
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = ret[i] - val[i];
        adv[i] = normalize ? (a - mean) / denom : a;
    }
}

__global__ __launch_bounds__(256) void ring_store_kernel(const float* __restrict__ src, int64_t ld_src, int D,
                                                         const int32_t* __restrict__ idx, int remap_h, int remap_n, int n,
                                                         float* __restrict__ dst, int64_t size, int64_t head) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r = blockIdx.x * 4 + ty;
    if (r >= n) return;
    const int64_t p = map_row(r, idx, remap_h, remap_n);
    const int64_t q = (head + r) % size;
    for (int j = tx; j < D; j += 64) dst[q * D + j] = src[p * ld_src + j];
}

// Philox4x32-10
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u;
    k[1] += 0xBB67AE85u;
}

__device__ __forceinline__ float philox_normal(uint64_t seed, uint64_t offset, uint64_t elem) {
    uint32_t c[4] = {(uint32_t)elem, (uint32_t)(elem >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int i = 0; i < 10; ++i) philox_round(c, k);
    const float u1 = ((float)c[0] + 1.0f) * 2.3283064365386963e-10f;  // (0, 1]
    const float u2 = (float)c[1] * 2.3283064365386963e-10f;
    return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__device__ __forceinline__ float philox_uniform(uint64_t seed, uint64_t offset, uint64_t elem) {
    uint32_t c[4] = {(uint32_t)elem, (uint32_t)(elem >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int i = 0; i < 10; ++i) philox_round(c, k);
    return (float)c[2] * 2.3283064365386963e-10f;                    // [0, 1)
}

// Rollout-time action head (learning/amp_models.py:29-36 eval branch + learning/amp_agent.py:160-166): one wave per row,
//   mu (optionally tanh, learning/hrl_network_builder.py:26-29) -> a = mu + sigma * N(0, 1) -> neglogp(a) -> eps-greedy:
//   rows whose Bernoulli(p_row) draw is 0 take the deterministic action mu (the stored neglogp stays the sampled one).
__global__ __launch_bounds__(256) void sample_actions_kernel(const float* __restrict__ mu, int64_t ld_mu,
                                                             const float* __restrict__ logstd,
                                                             const float* __restrict__ rand_probs,
                                                             const uint64_t* __restrict__ rng, float* __restrict__ mu_out,
                                                             float* __restrict__ sigma_out, float* __restrict__ actions,
                                                             float* __restrict__ neglogp, float* __restrict__ rand_mask,
                                                             int n, int A, int mu_tanh) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const uint64_t seed = rng[0], off = rng[1];
    float m = 0.f, s = 1.f, a = 0.f, q = 0.f, ls = 0.f;
    if (lane < A) {
        m = mu[(int64_t)r * ld_mu + lane];
        if (mu_tanh) m = tanhf(m);
        ls = logstd[lane];
        s = expf(ls);
        a = m + s * philox_normal(seed, off, (uint64_t)r * A + lane);
        const float d = (a - m) / s;
        q = d * d;
    }
    const float nlp = 0.5f * wave_sum(q) + 0.5f * 1.8378770664093453f * (float)A + wave_sum(ls);
    float keep = 1.f;
    if (rand_probs) keep = philox_uniform(seed, off, (uint64_t)n * A + r) < rand_probs[r] ? 1.f : 0.f;
    if (lane < A) {
        mu_out[(int64_t)r * A + lane] = m;
        sigma_out[(int64_t)r * A + lane] = s;
        actions[(int64_t)r * A + lane] = keep != 0.f ? a : m;
    }
    if (lane == 0) {
        neglogp[r] = nlp;
        if (rand_mask) rand_mask[r] = keep;
    }
}

// y[r, :] = x[r, :] / max(|x[r, :]|, 1e-12)   (torch.nn.functional.normalize, dim <= 128): one wave per row
__global__ __launch_bounds__(256) void normalize_rows_kernel(const float* __restrict__ x, int64_t ld_x, float* __restrict__ y,
                                                             int64_t ld_y, int n, int dim) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float v[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        if (j < dim) v[q] = x[(int64_t)r * ld_x + j];
    }
    const float nrm = fmaxf(sqrtf(wave_sum(v[0] * v[0] + v[1] * v[1])), 1e-12f);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        if (j < dim) y[(int64_t)r * ld_y + j] = v[q] / nrm;
    }
}

// one wave per row, dim <= 128
__global__ __launch_bounds__(256) void sample_latents_kernel(float* __restrict__ z, int rows, int dim,
                                                             const uint64_t* __restrict__ rng, int64_t row_offset,
                                                             void* __restrict__ z2, int64_t ld_z2, int z2_dtype) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const uint64_t seed = rng[0], off = rng[1];
    float v[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        if (j < dim) v[q] = philox_normal(seed, off, (uint64_t)(row_offset + r) * dim + j);
    }
    const float nrm = fmaxf(sqrtf(wave_sum(v[0] * v[0] + v[1] * v[1])), 1e-12f);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        if (j < dim) {
            const float o = v[q] / nrm;
            z[(int64_t)r * dim + j] = o;
            if (z2) {
                if (z2_dtype == ASE_BF16) reinterpret_cast<bf16_t*>(z2)[(int64_t)r * ld_z2 + j] = (bf16_t)o;
                else if (z2_dtype == ASE_F16) reinterpret_cast<f16_t*>(z2)[(int64_t)r * ld_z2 + j] = from_f32<f16_t>(o);
                else reinterpret_cast<float*>(z2)[(int64_t)r * ld_z2 + j] = o;
            }
        }
    }
}

__global__ void rng_advance_kernel(uint64_t* rng) { rng[1] += 1; }

inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int ase_hip_disc_reward(const float* logit, int64_t ld_l, float* r, int64_t n, float scale, void* stream) {
    ASE_CHECK_ARG(logit && r && n > 0, "disc_reward: null/empty operand");
    ASE_LAUNCH(disc_reward_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, logit, ld_l, r, n, scale);
    ASE_CHECK_LAUNCH("disc_reward");
    return ASE_OK;
}

extern "C" int ase_hip_enc_reward(const float* e, int64_t ld_e, const float* z, int64_t ld_z, float* r, int64_t n,
                                  int z_dim, float scale, void* stream) {
    ASE_CHECK_ARG(e && z && r && n > 0 && z_dim >= 1 && z_dim <= 128, "enc_reward: bad operand");
    ASE_LAUNCH(enc_reward_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, e, ld_e, z,
                       ld_z, r, n, z_dim, scale);
    ASE_CHECK_LAUNCH("enc_reward");
    return ASE_OK;
}

extern "C" int ase_hip_gae(const uint8_t* dones, const float* values, const float* next_values, const float* r_task,
                           const float* r_disc, const float* r_enc, float w_task, float w_disc, float w_enc,
                           double gamma, double tau, float* advs, float* returns, int H, int N, void* stream) {
    ASE_CHECK_ARG(dones && values && next_values && r_task && advs && returns && H > 0 && N > 0,
                  "gae: null/empty operand");
    ASE_LAUNCH(gae_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, dones, values, next_values,
                       r_task, r_disc, r_enc, w_task, w_disc, w_enc, (float)gamma, (float)(gamma * tau), advs, returns,
                       H, N);
    ASE_CHECK_LAUNCH("gae");
    return ASE_OK;
}

extern "C" int ase_hip_adv_norm(const float* returns, const float* values, const float* mask, float* adv, double* acc3,
                                int64_t n, int normalize, int phase, void* stream) {
    ASE_CHECK_ARG(returns && values && acc3 && n > 1, "adv_norm: null/empty operand");
    if (phase == 0) {
        ASE_LAUNCH(adv_moments_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, returns, values,
                           mask, acc3, n);
    } else {
        ASE_CHECK_ARG(adv, "adv_norm: null output");
        ASE_LAUNCH(adv_apply_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, returns, values, adv,
                           acc3, n, normalize, mask != nullptr);
    }
    ASE_CHECK_LAUNCH("adv_norm");
    return ASE_OK;
}

extern "C" int ase_hip_ring_store(const float* src, int64_t ld_src, int D, const int32_t* idx, int remap_h, int remap_n,
                                  int n, float* dst, int64_t size, int64_t head, void* stream) {
    ASE_CHECK_ARG(src && dst && D > 0 && n > 0 && n <= size && head >= 0 && head < size, "ring_store: bad operand");
    ASE_LAUNCH(ring_store_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, src, ld_src, D, idx,
                       remap_h, remap_n, n, dst, size, head);
    ASE_CHECK_LAUNCH("ring_store");
    return ASE_OK;
}

extern "C" int ase_hip_normalize_rows(const float* x, int64_t ld_x, float* y, int64_t ld_y, int n, int dim, void* stream) {
    ASE_CHECK_ARG(x && y && n > 0 && dim >= 1 && dim <= 128 && ld_x >= dim && ld_y >= dim, "normalize_rows: bad operand");
    ASE_LAUNCH(normalize_rows_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ld_x, y, ld_y, n, dim);
    ASE_CHECK_LAUNCH("normalize_rows");
    return ASE_OK;
}

extern "C" int ase_hip_sample_actions(const float* mu, int64_t ld_mu, const float* logstd, const float* rand_probs,
                                      uint64_t* rng_state, float* mu_out, float* sigma_out, float* actions,
                                      float* neglogp, float* rand_mask, int n, int act_dim, int mu_tanh, void* stream) {
    ASE_CHECK_ARG(mu && logstd && rng_state && mu_out && sigma_out && actions && neglogp && n > 0 && act_dim >= 1 &&
                      act_dim <= 64 && ld_mu >= act_dim, "sample_actions: bad operand");
    ASE_LAUNCH(sample_actions_kernel, dim3((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, mu, ld_mu, logstd,
                       rand_probs, rng_state, mu_out, sigma_out, actions, neglogp, rand_mask, n, act_dim, mu_tanh);
    ASE_LAUNCH(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state);
    ASE_CHECK_LAUNCH("sample_actions");
    return ASE_OK;
}

extern "C" int ase_hip_sample_latents(float* z, int rows, int dim, uint64_t* rng_state, int64_t row_offset, int advance,
                                      void* z2, int64_t ld_z2, int z2_dtype, void* stream) {
    ASE_CHECK_ARG(z && rng_state && rows > 0 && dim >= 1 && dim <= 128 && row_offset >= 0, "sample_latents: bad operand");
    ASE_CHECK_ARG(z2 == nullptr || ((z2_dtype == ASE_BF16 || z2_dtype == ASE_F32 || z2_dtype == ASE_F16) && ld_z2 >= dim), "sample_latents: bad second output");
    ASE_LAUNCH(sample_latents_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, z, rows, dim,
               rng_state, row_offset, z2, ld_z2, z2_dtype);
    if (advance) ASE_LAUNCH(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state);
    ASE_CHECK_LAUNCH("sample_latents");
    return ASE_OK;
}
