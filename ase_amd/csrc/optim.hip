// Optimizer step (torch.optim.Adam semantics) and the weight-only gradient terms.
#include "common.h"

namespace {

// opt_state (f64[8]): {step, lr, beta1, beta2, eps, bias_corr1, bias_corr2, unused}
__global__ void begin_step_kernel(double* __restrict__ opt_state, double* __restrict__ acc, int n_acc,
                                  double* __restrict__ zero2, int n_zero2, unsigned long long* __restrict__ rng_bump) {
    for (int i = threadIdx.x; i < n_acc; i += blockDim.x) acc[i] = 0.0;
    for (int i = threadIdx.x; i < n_zero2; i += blockDim.x) zero2[i] = 0.0;
    if (threadIdx.x == 0 && rng_bump) rng_bump[1] += 1;
    if (threadIdx.x == 0 && opt_state) {
        const double step = opt_state[0] + 1.0;
        opt_state[0] = step;
        opt_state[5] = 1.0 - pow(opt_state[2], step);
        opt_state[6] = 1.0 - pow(opt_state[3], step);
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   const double* __restrict__ st) {
    // scalars are formed in double (as torch does in Python) and applied in f32
    const float one_m_b1 = (float)(1.0 - st[2]);
    const float b2 = (float)st[3], one_m_b2 = (float)(1.0 - st[3]);
    const float eps = (float)st[4];
    const float step_size = (float)(st[1] / st[5]);
    const float bc2_sqrt = (float)sqrt(st[6]);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        float mi = m[i], vi = v[i];
        mi = mi + one_m_b1 * (gi - mi);                 // exp_avg.lerp_(grad, 1 - beta1)
        vi = vi * b2;                                   // exp_avg_sq.mul_(beta2)
        vi = vi + one_m_b2 * (gi * gi);                 //            .addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        w[i] = w[i] - step_size * (mi / denom);         // param.addcdiv_(exp_avg, denom, value=-step_size)
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ __launch_bounds__(256) void axpy_kernel(float* __restrict__ g, const float* __restrict__ w, int64_t n, float c) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        g[i] = g[i] + c * w[i];
}

// ---- fused optimizer step over every dense layer (one launch) -------------------------------------------------------
// desc[l] (int64 x 24), one entry per weight matrix:
//   0 W  1 n_real  2 k_real  3 Ws  4 ldws  5 Wts  6 ldwts  7 split_src  8 gap  9 bias  10 bias_shadow  11 tiles_k
//   12 gW  13 mW  14 vW  15 gb  16 mb  17 vb  18 wd coefficient (f32 bits)  19 acc slot A (or -1)  20 acc slot B (or -1)
//   21 wide (1: k_real, both shadow pitches, split_src and gap are multiples of 4, shadows 16-byte aligned -> 16-byte path)
// Per 32 x 32 tile of W (rows coalesced): g += wd * w (weight-only loss terms), sum of w^2 of the PRE-update weights
// into acc[slot A / B] (reported regularisers), Adam, then the fresh weight goes straight into the compute-dtype
// shadows W_s (row-major) and W_s^T (through an LDS transpose).  The first workgroup of a layer also steps the bias.
__device__ __forceinline__ float adam_elem(float w, float gi, float& mi, float& vi, float one_m_b1, float b2,
                                           float one_m_b2, float eps, float step_size, float bc2_sqrt) {
    mi = mi + one_m_b1 * (gi - mi);                 // exp_avg.lerp_(grad, 1 - beta1)
    vi = vi * b2;                                   // exp_avg_sq.mul_(beta2)
    vi = vi + one_m_b2 * (gi * gi);                 //            .addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    return w - step_size * (mi / denom);            // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// 16-byte accesses on the flat f32 buffers (parameters / gradients / moments start at arbitrary 4-byte offsets behind the
// odd-sized bias vectors: dword alignment is all the hardware asks of a global dwordx4)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

template <typename T> __device__ __forceinline__ void store_shadow4(T* p, const float (&w)[4]);
template <> __device__ __forceinline__ void store_shadow4<bf16_t>(bf16_t* p, const float (&w)[4]) {
    bf16x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = (bf16_t)w[c];
    *reinterpret_cast<bf16x4*>(p) = v;
}
template <> __device__ __forceinline__ void store_shadow4<f16_t>(f16_t* p, const float (&w)[4]) {
    f16x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = from_f32<f16_t>(w[c]);
    *reinterpret_cast<f16x4*>(p) = v;
}
template <> __device__ __forceinline__ void store_shadow4<float>(float* p, const float (&w)[4]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{w[0], w[1], w[2], w[3]};
}

template <typename T>
__global__ __launch_bounds__(256) void apply_multi_kernel(const int64_t* __restrict__ desc, const double* __restrict__ st,
                                                          double* __restrict__ acc) {
    __shared__ float tile[32][129];          // scalar path: [32][33] of it
    __shared__ double red[16];
    const int64_t* d = desc + 24 * blockIdx.y;
    float* W = reinterpret_cast<float*>(d[0]);
    const int n_real = (int)d[1], k_real = (int)d[2];
    T* Ws = reinterpret_cast<T*>(d[3]);
    const int64_t ldws = d[4];
    T* Wts = reinterpret_cast<T*>(d[5]);
    const int64_t ldwts = d[6];
    const int split_src = (int)d[7], gap = (int)d[8];
    const int tiles_k = (int)d[11];
    float* gW = reinterpret_cast<float*>(d[12]);
    float* mW = reinterpret_cast<float*>(d[13]);
    float* vW = reinterpret_cast<float*>(d[14]);
    const float wd = __builtin_bit_cast(float, (int)d[18]);
    const int slot_a = (int)d[19], slot_b = (int)d[20];
    const bool adam = st != nullptr;
    float one_m_b1 = 0.f, b2 = 0.f, one_m_b2 = 0.f, eps = 0.f, step_size = 0.f, bc2_sqrt = 1.f;
    if (adam) {      // scalars are formed in double (as torch does in Python) and applied in f32
        one_m_b1 = (float)(1.0 - st[2]); b2 = (float)st[3]; one_m_b2 = (float)(1.0 - st[3]);
        eps = (float)st[4]; step_size = (float)(st[1] / st[5]); bc2_sqrt = (float)sqrt(st[6]);
    }
    const int tiles = tiles_k * ((n_real + 31) / 32);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (blockIdx.x == 0 && d[9]) {
        float* b = reinterpret_cast<float*>(d[9]);
        float* bs = reinterpret_cast<float*>(d[10]);
        float* gb = reinterpret_cast<float*>(d[15]);
        float* mb = reinterpret_cast<float*>(d[16]);
        float* vb = reinterpret_cast<float*>(d[17]);
        for (int i = threadIdx.x; i < n_real; i += 256) {
            float w = b[i];
            if (adam) {
                float mi = mb[i], vi = vb[i];
                w = adam_elem(w, gb[i], mi, vi, one_m_b1, b2, one_m_b2, eps, step_size, bc2_sqrt);
                b[i] = w; mb[i] = mi; vb[i] = vi;
            }
            bs[i] = w;
        }
    }
    double w2[1] = {0.0};
    // ---- wide path (desc[21]: rows in whole 16-byte chunks on every side): 32 x 128 tiles, one 16-byte access per array
    // and thread for 4 weights, 16 transposed shadow values (32 / 64 bytes) per store instead of one
    if (d[21]) {
        const int tk = (k_real + 127) / 128;
        const int tilesv = tk * ((n_real + 31) / 32);
        const int vx = threadIdx.x & 31, vy = threadIdx.x >> 5;
        for (int t = blockIdx.x; t < tilesv; t += gridDim.x) {
            const int k0 = (t % tk) * 128, n0 = (t / tk) * 32;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + vy + 8 * i, k = k0 + 4 * vx;
                float w[4] = {0.f, 0.f, 0.f, 0.f};
                if (n < n_real && k < k_real) {
                    const int64_t o = (int64_t)n * k_real + k;
                    const f4u wv = *reinterpret_cast<const f4u*>(W + o);
#pragma unroll
                    for (int c = 0; c < 4; ++c) w[c] = wv[c];
                    if (adam) {
                        f4u gv = *reinterpret_cast<const f4u*>(gW + o);
                        f4u mv = *reinterpret_cast<const f4u*>(mW + o), vv = *reinterpret_cast<const f4u*>(vW + o);
                        if (slot_a >= 0) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) w2[0] += (double)w[c] * (double)w[c];
                        }
                        if (wd != 0.f) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) gv[c] = gv[c] + wd * w[c];
                            *reinterpret_cast<f4u*>(gW + o) = gv;      // the exported gradient includes the weight-only terms
                        }
                        f4u wn;
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float mi = mv[c], vi = vv[c];
                            w[c] = adam_elem(w[c], gv[c], mi, vi, one_m_b1, b2, one_m_b2, eps, step_size, bc2_sqrt);
                            mv[c] = mi; vv[c] = vi; wn[c] = w[c];
                        }
                        *reinterpret_cast<f4u*>(W + o) = wn;
                        *reinterpret_cast<f4u*>(mW + o) = mv;
                        *reinterpret_cast<f4u*>(vW + o) = vv;
                    }
                    store_shadow4<T>(Ws + (int64_t)n * ldws + ((k < split_src) ? k : k + gap), w);
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) tile[vy + 8 * i][4 * vx + c] = w[c];
            }
            __syncthreads();
            const int kk = threadIdx.x >> 1, half = threadIdx.x & 1;
            const int k = k0 + kk, nb = n0 + 16 * half;
            if (k < k_real && nb < n_real) {
                T* dst = Wts + (int64_t)((k < split_src) ? k : k + gap) * ldwts + nb;
                if (nb + 16 <= n_real) {
#pragma unroll
                    for (int q = 0; q < 16; q += 4) {
                        const float w4[4] = {tile[16 * half + q][kk], tile[16 * half + q + 1][kk], tile[16 * half + q + 2][kk],
                                             tile[16 * half + q + 3][kk]};
                        store_shadow4<T>(dst + q, w4);
                    }
                } else {
                    for (int q = 0; q < 16 && nb + q < n_real; ++q) dst[q] = from_f32<T>(tile[16 * half + q][kk]);
                }
            }
        }
    } else
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int k0 = (t % tiles_k) * 32, n0 = (t / tiles_k) * 32;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + ty + 8 * i, k = k0 + tx;
            float w = 0.f;
            if (n < n_real && k < k_real) {
                const int64_t o = (int64_t)n * k_real + k;
                w = W[o];
                if (adam) {
                    if (slot_a >= 0) w2[0] += (double)w * (double)w;
                    float gi = gW[o];
                    if (wd != 0.f) {
                        gi = gi + wd * w;
                        gW[o] = gi;                       // the exported gradient includes the weight-only terms
                    }
                    float mi = mW[o], vi = vW[o];
                    w = adam_elem(w, gi, mi, vi, one_m_b1, b2, one_m_b2, eps, step_size, bc2_sqrt);
                    W[o] = w; mW[o] = mi; vW[o] = vi;
                }
                Ws[(int64_t)n * ldws + ((k < split_src) ? k : k + gap)] = from_f32<T>(w);
            }
            tile[ty + 8 * i][tx] = w;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ty + 8 * i, n = n0 + tx;
            if (k < k_real && n < n_real)
                Wts[(int64_t)((k < split_src) ? k : k + gap) * ldwts + n] = from_f32<T>(tile[tx][ty + 8 * i]);
        }
    }
    if (adam && slot_a >= 0) {
        __syncthreads();
        block_sum<1>(w2, red);
        if (threadIdx.x == 0 && w2[0] != 0.0) {
            atomic_add_f64(acc + slot_a, w2[0]);
            if (slot_b >= 0) atomic_add_f64(acc + slot_b, w2[0]);
        }
    }
}

inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

// g *= min(1, max_norm / (sqrt(sqnorm) + 1e-6)): torch.nn.utils.clip_grad_norm_ with the squared total norm already on the
// device (learning/ase_agent.py:273-288, the truncate_grads branch)
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, int64_t n, const double* __restrict__ sqnorm,
                                                         float max_norm) {
    const float total = (float)sqrt(*sqnorm);
    const float coef = fminf(max_norm / (total + 1e-6f), 1.0f);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] *= coef;
}

}  // namespace

extern "C" int ase_hip_clip_scale(float* g, int64_t n, const double* sqnorm, float max_norm, void* stream) {
    ASE_CHECK_ARG(g && sqnorm && n > 0 && max_norm > 0.f, "clip_scale: bad operand");
    ASE_LAUNCH(clip_scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, n, sqnorm, max_norm);
    ASE_CHECK_LAUNCH("clip_scale");
    return ASE_OK;
}

extern "C" int ase_hip_begin_step(double* opt_state, double* acc, int n_acc, double* zero2, int n_zero2,
                                  uint64_t* rng_bump, void* stream) {
    ASE_CHECK_ARG((opt_state || acc || zero2 || rng_bump) && n_acc >= 0 && n_zero2 >= 0, "begin_step: nothing to do");
    ASE_LAUNCH(begin_step_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, opt_state, acc, acc ? n_acc : 0, zero2,
                       zero2 ? n_zero2 : 0, (unsigned long long*)rng_bump);
    ASE_CHECK_LAUNCH("begin_step");
    return ASE_OK;
}

extern "C" int ase_hip_adam(float* w, const float* g, float* m, float* v, int64_t n, const double* opt_state,
                            void* stream) {
    ASE_CHECK_ARG(w && g && m && v && opt_state && n > 0, "adam: null/empty operand");
    ASE_LAUNCH(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, opt_state);
    ASE_CHECK_LAUNCH("adam");
    return ASE_OK;
}

extern "C" int ase_hip_axpy(float* g, const float* w, int64_t n, float c, void* stream) {
    ASE_CHECK_ARG(g && w && n > 0, "axpy: null/empty operand");
    ASE_LAUNCH(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, w, n, c);
    ASE_CHECK_LAUNCH("axpy");
    return ASE_OK;
}

extern "C" int ase_hip_apply_multi(const int64_t* desc, int n_layers, const double* opt_state, double* acc, int dtype,
                                   void* stream) {
    ASE_CHECK_ARG(desc && n_layers > 0, "apply_multi: null/empty operand");
    ASE_CHECK_ARG(opt_state == nullptr || acc != nullptr, "apply_multi: optimizer step without the accumulator array");
    const dim3 grid(256, n_layers);
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH(apply_multi_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, desc, opt_state, acc);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "apply_multi: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("apply_multi");
    return ASE_OK;
}
