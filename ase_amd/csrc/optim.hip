// Optimizer step (torch.optim.Adam semantics) and the weight-only gradient terms.
#include "common.h"

namespace {

// opt_state (f64[8]): {step, lr, beta1, beta2, eps, bias_corr1, bias_corr2, unused}
__global__ void begin_step_kernel(double* __restrict__ opt_state, double* __restrict__ acc, int n_acc) {
    for (int i = threadIdx.x; i < n_acc; i += blockDim.x) acc[i] = 0.0;
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

inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

}  // namespace

extern "C" int ase_hip_begin_step(double* opt_state, double* acc, int n_acc, void* stream) {
    ASE_CHECK_ARG((opt_state || acc) && n_acc >= 0, "begin_step: nothing to do");
    hipLaunchKernelGGL(begin_step_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, opt_state, acc, acc ? n_acc : 0);
    ASE_CHECK_LAUNCH("begin_step");
    return ASE_OK;
}

extern "C" int ase_hip_adam(float* w, const float* g, float* m, float* v, int64_t n, const double* opt_state,
                            void* stream) {
    ASE_CHECK_ARG(w && g && m && v && opt_state && n > 0, "adam: null/empty operand");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, w, g, m, v, n, opt_state);
    ASE_CHECK_LAUNCH("adam");
    return ASE_OK;
}

extern "C" int ase_hip_axpy(float* g, const float* w, int64_t n, float c, void* stream) {
    ASE_CHECK_ARG(g && w && n > 0, "axpy: null/empty operand");
    hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, g, w, n, c);
    ASE_CHECK_LAUNCH("axpy");
    return ASE_OK;
}
