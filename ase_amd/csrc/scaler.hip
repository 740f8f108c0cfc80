// Loss scaler of the half-storage engine: what torch.cuda.amp.GradScaler does around the optimizer step in the reference's
// mixed_precision path (learning/ase_agent.py:271-288, learning/amp_agent.py:354-371, learning/common_agent.py:417-418) -
// found_inf over everything the scaled backward produced, a SKIPPED optimizer step when it fired, and GradScaler.update() - backoff
// after a skipped step, growth after growth_interval clean ones - once per optimisation step, on the device (ABI 6).
//   The engine's conversions into half storage saturate at +-65504 (common.h from_f32<f16_t>: an overflowing scaled gradient must
// not become inf -> NaN inside the matrix launches that follow), so "overflow" here = an element that is not finite OR sits at
// half's saturation value.  The scale lives in the scaler state; scaler_book_kernel rewrites the table of scale RECORDS
// {S, n}, {1 / S, n}, {1 / S^2, n}, {1, n} the loss heads, the matrix launches and the penalty's norm get through their `*_dev`
// arguments, so recorded launch programs survive every change of the scale - and the same launches report what they stored into the
// record's count n (common.h ovf_report): the producers detect, ase_hip_scaler_check(_multi) is left for buffers whose writers were
// given no record and for the f32 gradient.  (scale_tab NULL: the ABI-5 form - the scale as a launch argument the host moves
// between updates, detection by check launches only.)
// Own translation unit: nothing of the static-scale path links against it.
#include "common.h"

namespace {

// scaler (f64[8]): {found (elements / workgroups that overflowed since the last scaler_step), skipped steps (total), GradScaler's
//                   growth tracker (clean steps since the scale last moved), steps seen (total), scale, growth_factor, backoff_factor,
//                   growth_interval}
enum { SC_FOUND = 0, SC_SKIPPED = 1, SC_CLEAN = 2, SC_STEPS = 3, SC_SCALE = 4, SC_GROWTH = 5, SC_BACKOFF = 6, SC_INTERVAL = 7 };

template <typename T> __device__ __forceinline__ bool overflowed(T x);
template <> __device__ __forceinline__ bool overflowed<float>(float x) { return !(fabsf(x) <= 3.402823466e38f); }      // NaN, +-inf
template <> __device__ __forceinline__ bool overflowed<bf16_t>(bf16_t x) { return !(fabsf((float)x) <= 3.402823466e38f); }
template <> __device__ __forceinline__ bool overflowed<f16_t>(f16_t x) { return !(fabsf((float)x) < 65504.f); }        // + saturated

// 16-byte loads over the aligned body, scalar loads over head and tail; one f64 atomic per workgroup that found something
// (bx / nbx: this workgroup's place among the workgroups that share the buffer)
template <typename T>
__device__ __forceinline__ void scaler_check_body(const T* __restrict__ x, int64_t n, double* __restrict__ scaler, int bx, int nbx) {
    constexpr int V = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    __shared__ int any;
    if (threadIdx.x == 0) any = 0;
    __syncthreads();
    // elements in front of the first 16-byte boundary (buffers of the engine start aligned: head = 0 there)
    const uintptr_t addr = reinterpret_cast<uintptr_t>(x);
    int64_t head = (int64_t)(((16 - (addr & 15)) & 15) / sizeof(T));
    if (head > n) head = n;
    const int64_t nvec = (n - head) / V;
    const vec_t* xv = reinterpret_cast<const vec_t*>(x + head);
    bool bad = false;
    const int64_t tid = bx * (int64_t)blockDim.x + threadIdx.x, nthr = (int64_t)nbx * blockDim.x;
    for (int64_t i = tid; i < nvec; i += nthr) {
        const vec_t v = xv[i];
#pragma unroll
        for (int c = 0; c < V; ++c) bad |= overflowed<T>(v[c]);
    }
    for (int64_t i = tid; i < head; i += nthr) bad |= overflowed<T>(x[i]);
    for (int64_t i = head + nvec * V + tid; i < n; i += nthr) bad |= overflowed<T>(x[i]);
    if (bad) any = 1;                          // (every writer stores the same value)
    __syncthreads();
    if (threadIdx.x == 0 && any) atomic_add_f64(scaler + SC_FOUND, 1.0);
}

template <typename T>
__global__ __launch_bounds__(256) void scaler_check_kernel(const T* __restrict__ x, int64_t n, double* __restrict__ scaler) {
    scaler_check_body<T>(x, n, scaler, blockIdx.x, gridDim.x);
}

// one launch over a table of buffers: row b = {pointer, elements, storage type}; gridDim.y = rows, gridDim.x workgroups share a buffer
// (a workgroup whose share of a small buffer is empty leaves at once)
__global__ __launch_bounds__(256) void scaler_check_multi_kernel(const int64_t* __restrict__ table, double* __restrict__ scaler) {
    const int64_t* row = table + 3 * (int64_t)blockIdx.y;
    const int64_t n = row[1];
    const int dtype = (int)row[2];
    const int64_t per_wg = 256 * (16 / (dtype == ASE_F32 ? 4 : 2));
    if ((int64_t)blockIdx.x * per_wg >= n) return;
    if (dtype == ASE_F32) scaler_check_body<float>(reinterpret_cast<const float*>(row[0]), n, scaler, blockIdx.x, gridDim.x);
    else if (dtype == ASE_BF16) scaler_check_body<bf16_t>(reinterpret_cast<const bf16_t*>(row[0]), n, scaler, blockIdx.x, gridDim.x);
    else scaler_check_body<f16_t>(reinterpret_cast<const f16_t*>(row[0]), n, scaler, blockIdx.x, gridDim.x);
}

// what the producers reported into the records' counts (scale_tab[2 k + 1])
__device__ __forceinline__ float tab_found(const float* __restrict__ scale_tab) {
    return scale_tab ? scale_tab[1] + scale_tab[3] + scale_tab[5] + scale_tab[7] : 0.f;
}

// records -> scaler[found] (data parallel: the ranks SUM-exchange that one number between this launch and ase_hip_scaler_step)
__global__ void scaler_fold_kernel(double* __restrict__ scaler, float* __restrict__ scale_tab) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    scaler[SC_FOUND] += (double)tab_found(scale_tab);
    for (int k = 0; k < 4; ++k) scale_tab[2 * k + 1] = 0.f;
}

// found: the step's gradient is dropped (the optimizer launch behind this one then runs the identity step scaler_book writes)
__global__ __launch_bounds__(256) void scaler_guard_kernel(float* __restrict__ g, int64_t n, const double* __restrict__ scaler,
                                                           const float* __restrict__ scale_tab) {
    if (scaler[SC_FOUND] == 0.0 && tab_found(scale_tab) == 0.f) return;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) g[i] = 0.f;
}

// opt_state (f64[8]): {step, lr, beta1, beta2, eps, bias_corr1, bias_corr2, unused} (optim.hip)
__global__ void scaler_book_kernel(double* __restrict__ scaler, double* __restrict__ opt_state, double* __restrict__ opt_eff,
                                   float* __restrict__ scale_tab) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool found = scaler[SC_FOUND] != 0.0 || tab_found(scale_tab) != 0.f;
    if (found) {
        // GradScaler.step does not call optimizer.step(): the step counter begin_step advanced goes back (its bias corrections
        // are recomputed from the counter by the next begin_step), and the optimizer launch gets the identity step -
        // exp_avg.lerp_(g, 0), exp_avg_sq * 1 + 0 * g^2, param - 0 * (...) - on the zeroed gradient
        opt_state[0] -= 1.0;
        opt_eff[0] = opt_state[0];
        opt_eff[1] = 0.0;                      // lr
        opt_eff[2] = 1.0;                      // beta1
        opt_eff[3] = 1.0;                      // beta2
        opt_eff[4] = opt_state[4];             // eps (> 0 keeps the quotient finite)
        opt_eff[5] = 1.0;
        opt_eff[6] = 1.0;
        opt_eff[7] = opt_state[7];
        scaler[SC_SKIPPED] += 1.0;
        scaler[SC_CLEAN] = 0.0;
    } else {
        for (int i = 0; i < 8; ++i) opt_eff[i] = opt_state[i];
        scaler[SC_CLEAN] += 1.0;
    }
    scaler[SC_STEPS] += 1.0;
    scaler[SC_FOUND] = 0.0;
    if (scale_tab) {
        // torch/amp/grad_scaler.py _amp_update_scale_: found -> scale *= backoff_factor, tracker = 0; else tracker += 1 (above) and at
        // growth_interval: scale *= growth_factor, tracker = 0
        double s = scaler[SC_SCALE];
        if (found) s *= scaler[SC_BACKOFF];
        else if (scaler[SC_CLEAN] >= scaler[SC_INTERVAL]) {
            s *= scaler[SC_GROWTH];
            scaler[SC_CLEAN] = 0.0;
        }
        scaler[SC_SCALE] = s;
        scale_tab[0] = (float)s;
        scale_tab[2] = (float)(1.0 / s);
        scale_tab[4] = (float)(1.0 / (s * s));
        scale_tab[6] = 1.f;
        for (int k = 0; k < 4; ++k) scale_tab[2 * k + 1] = 0.f;
    }
}

inline int check_grid(int64_t n, int elem) {
    const int64_t per_wg = 256 * (16 / elem) * 4;          // four 16-byte loads per thread
    int64_t g = (n + per_wg - 1) / per_wg;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace

extern "C" int ase_hip_scaler_check(const void* buf, int64_t n, int dtype, double* scaler, void* stream) {
    ASE_CHECK_ARG(buf && scaler && n > 0, "scaler_check: null/empty operand");
    ASE_CHECK_ARG(dtype == ASE_F32 || dtype == ASE_BF16 || dtype == ASE_F16, "scaler_check: bad dtype %d", dtype);
    ASE_CHECK_ARG((reinterpret_cast<uintptr_t>(buf) % ase_elem_size(dtype)) == 0, "scaler_check: misaligned buffer");
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH(scaler_check_kernel<T>, dim3(check_grid(n, (int)sizeof(T))), dim3(256), 0, (hipStream_t)stream,
                   static_cast<const T*>(buf), n, scaler);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "scaler_check: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("scaler_check");
    return ASE_OK;
}

extern "C" int ase_hip_scaler_check_multi(const int64_t* table, int n_bufs, int wg_per_buf, double* scaler, void* stream) {
    ASE_CHECK_ARG(table && scaler && n_bufs > 0 && n_bufs <= 65535, "scaler_check_multi: null/empty table");
    ASE_CHECK_ARG(wg_per_buf >= 1 && wg_per_buf <= 4096, "scaler_check_multi: wg_per_buf %d not in [1, 4096]", wg_per_buf);
    ASE_LAUNCH(scaler_check_multi_kernel, dim3(wg_per_buf, n_bufs), dim3(256), 0, (hipStream_t)stream, table, scaler);
    ASE_CHECK_LAUNCH("scaler_check_multi");
    return ASE_OK;
}

extern "C" int ase_hip_scaler_fold(double* scaler, float* scale_tab, void* stream) {
    ASE_CHECK_ARG(scaler && scale_tab, "scaler_fold: null operand");
    ASE_LAUNCH(scaler_fold_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler, scale_tab);
    ASE_CHECK_LAUNCH("scaler_fold");
    return ASE_OK;
}

extern "C" int ase_hip_scaler_step(double* scaler, double* opt_state, double* opt_eff, float* grads, int64_t n, float* scale_tab,
                                   void* stream) {
    ASE_CHECK_ARG(scaler && opt_state && opt_eff && grads && n > 0 && opt_eff != opt_state, "scaler_step: null/empty/aliased operand");
    int64_t g = (n + 1023) / 1024;
    g = g < 1 ? 1 : (g > 4096 ? 4096 : g);
    ASE_LAUNCH(scaler_guard_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, grads, n, (const double*)scaler,
               (const float*)scale_tab);
    ASE_LAUNCH(scaler_book_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scaler, opt_state, opt_eff, scale_tab);
    ASE_CHECK_LAUNCH("scaler_step");
    return ASE_OK;
}
