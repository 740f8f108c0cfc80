// Running mean/std input normaliser (rl_games RunningMeanStd semantics) fused with the minibatch
// row gather, plus the generic row gather/cast.  HBM-bound: every element is read once for the
// moments and once for the normalised write; nothing is materialised in between.
#include "common.h"

namespace {

// ---- batch moments: grid (col tiles of 64, row chunks); block 64 x 4 ---------------------------
constexpr int kRowsPerBlock = 64;

// 16-byte variant (D % 4 == 0, 16-byte aligned rows): a thread owns 4 columns; same row partition (rows = ty mod 4, ascending)
// and the same f64 arithmetic per element as the scalar (multi-stream) kernel below, so the sums are identical
__global__ __launch_bounds__(256) void rms_moments4_kernel(const float* __restrict__ src, int64_t ld_src, int D,
                                                           const int32_t* __restrict__ idx, int remap_h, int remap_n,
                                                           int M, const double* __restrict__ state,
                                                           double* __restrict__ sums) {
    __shared__ double red[3][8][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = (blockIdx.x * 64 + tx) * 4;
    const int r0 = blockIdx.y * kRowsPerBlock;
    const int r1 = min(M, r0 + kRowsPerBlock);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (j < D) {
        float shift[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) shift[c] = (float)state[j + c];
        for (int r = r0 + ty; r < r1; r += 16) {
            f32x4 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = r + 4 * q;
                if (rr < r1) x[q] = *reinterpret_cast<const f32x4*>(src + map_row(rr, idx, remap_h, remap_n) * ld_src + j);
                else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) x[q][c] = shift[c];
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const double d = (double)(x[q][c] - shift[c]);
                    s1[c] += d;
                    s2[c] += d * d;
                }
        }
    }
    if (ty > 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            red[ty - 1][c][tx] = s1[c];
            red[ty - 1][4 + c][tx] = s2[c];
        }
    }
    __syncthreads();
    if (ty == 0 && j < D) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // same association as the scalar kernel: ((t0 + t1) + t2) + t3
            const double a = ((s1[c] + red[0][c][tx]) + red[1][c][tx]) + red[2][c][tx];
            const double b = ((s2[c] + red[0][4 + c][tx]) + red[1][4 + c][tx]) + red[2][4 + c][tx];
            atomic_add_f64(sums + j + c, a);
            atomic_add_f64(sums + D + j + c, b);
        }
    }
}

// ---- up to 4 streams of the same width in ONE launch (agent / replay / demo AMP observations): grid z = stream.
// Same row partition and summation order as rms_moments_kernel (identical sums); the 16 row indices of a thread are
// fetched first and the 16 row loads are then all in flight at once (the gather makes each row a dependent index -> data
// chain: with 4 in flight the kernel ran at 1.4 TB/s).
struct RmsStreams {
    const float* src[4]; int64_t ld[4]; const int32_t* idx[4]; int rh[4], rn[4];
    double* sums[4];                       // moments
    const float* mean[4]; const float* stdv[4]; void* out[4]; int64_t ld_out[4];   // normalize
};

__global__ __launch_bounds__(256) void rms_moments_multi_kernel(RmsStreams S, int D, int M, const double* __restrict__ state) {
    __shared__ double red[2][4][64];
    const int s = blockIdx.z;
    const float* __restrict__ src = S.src[s];
    const int32_t* __restrict__ idx = S.idx[s];
    const int64_t ld_src = S.ld[s];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + tx;
    const int r0 = blockIdx.y * kRowsPerBlock;
    const int r1 = min(M, r0 + kRowsPerBlock);
    double s1 = 0.0, s2 = 0.0;
    if (j < D) {
        const float shift = (float)state[j];
        int64_t p[16];
        float x[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int rr = r0 + ty + 4 * q;
            p[q] = (rr < r1) ? map_row(rr, idx, S.rh[s], S.rn[s]) : -1;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) x[q] = (p[q] >= 0) ? src[p[q] * ld_src + j] : shift;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const double d = (double)(x[q] - shift);
            s1 += d;
            s2 += d * d;
        }
    }
    red[0][ty][tx] = s1;
    red[1][ty][tx] = s2;
    __syncthreads();
    if (ty == 0 && j < D) {
        s1 = red[0][0][tx] + red[0][1][tx] + red[0][2][tx] + red[0][3][tx];
        s2 = red[1][0][tx] + red[1][1][tx] + red[1][2][tx] + red[1][3][tx];
        atomic_add_f64(S.sums[s] + j, s1);
        atomic_add_f64(S.sums[s] + D + j, s2);
    }
}

// normalise up to 4 streams (one output each) in ONE launch: grid (col tiles, rows / 4, stream); 16-byte accesses
template <typename T>
__global__ __launch_bounds__(256) void rms_normalize_multi_kernel(RmsStreams S, int D, int M) {
    const int s = blockIdx.z;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r = blockIdx.y * 4 + ty;
    const int j = (blockIdx.x * 64 + tx) * 4;
    if (r >= M || j >= D) return;
    const int64_t p = map_row(r, S.idx[s], S.rh[s], S.rn[s]);
    const f32x4 x = *reinterpret_cast<const f32x4*>(S.src[s] + p * S.ld[s] + j);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(S.mean[s] + j), sd = *reinterpret_cast<const f32x4*>(S.stdv[s] + j);
    f32x4 y;
#pragma unroll
    for (int c = 0; c < 4; ++c) y[c] = fminf(fmaxf((x[c] - mu[c]) / sd[c], -5.f), 5.f);
    if constexpr (sizeof(T) == 2) {
        typename V16<T>::x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = from_f32<T>(y[c]);
        *reinterpret_cast<typename V16<T>::x4*>(reinterpret_cast<T*>(S.out[s]) + (int64_t)r * S.ld_out[s] + j) = v;
    } else {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(S.out[s]) + (int64_t)r * S.ld_out[s] + j) = y;
    }
}

// ---- merge into the running state; single block so `count` is read before it is rewritten ------
__global__ __launch_bounds__(1024) void rms_finalize_kernel(double* __restrict__ state, int D,
                                                           const double* __restrict__ sums, int count,
                                                           int n_streams, float* __restrict__ mean_out,
                                                           float* __restrict__ std_out) {
    const double count0 = state[2 * D];
    __syncthreads();
    for (int j = threadIdx.x; j < D; j += blockDim.x) {
        double mean = state[j], var = state[D + j], cnt = count0;
        const float shift = (float)mean;  // the shift rms_moments used (state is untouched in between)
        if (n_streams == 0) {
            mean_out[j] = (float)mean;
            std_out[j] = sqrtf((float)var + 1e-5f);
        }
        for (int s = 0; s < n_streams; ++s) {
            const double n = (double)count;
            const double s1 = sums[(int64_t)s * 2 * D + j], s2 = sums[(int64_t)s * 2 * D + D + j];
            // batch mean / unbiased variance, rounded to f32 like torch's x.mean(0) / x.var(0) on f32 data
            const double bm = (double)(float)((double)shift + s1 / n);
            const double bv = (double)(float)((s2 - s1 * s1 / n) / (n - 1.0));
            const double delta = bm - mean;
            const double tot = cnt + n;
            const double new_mean = mean + delta * n / tot;
            const double m2 = var * cnt + bv * n + delta * delta * cnt * n / tot;
            mean = new_mean;
            var = m2 / tot;
            cnt = tot;
            mean_out[(int64_t)s * D + j] = (float)mean;
            std_out[(int64_t)s * D + j] = sqrtf((float)var + 1e-5f);
        }
        state[j] = mean;
        state[D + j] = var;
    }
    if (threadIdx.x == 0 && n_streams > 0) state[2 * D] = count0 + (double)n_streams * (double)count;
}

template <typename T> __device__ __forceinline__ void store_opt(void* base, int64_t ld, int r, int j, float v) {
    if (base) reinterpret_cast<T*>(base)[(int64_t)r * ld + j] = from_f32<T>(v);
}

// grid (col tiles of 256, rows/4); block 256 = 64 cols x 4 rows  -> coalesced 256-B row segments
template <typename T>
__global__ __launch_bounds__(256) void rms_normalize_kernel(const float* __restrict__ src, int64_t ld_src, int D,
                                                            const int32_t* __restrict__ idx, int remap_h, int remap_n,
                                                            int M, const float* __restrict__ mean,
                                                            const float* __restrict__ stdv, void* out0, int64_t ld0,
                                                            void* out1, int64_t ld1, void* out2, int64_t ld2) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r = blockIdx.y * 4 + ty;
    if (r >= M) return;
    const int64_t p = map_row(r, idx, remap_h, remap_n);
    const float* row = src + p * ld_src;
    for (int j = blockIdx.x * 256 + tx; j < min(D, (int)(blockIdx.x + 1) * 256); j += 64) {
        float y = (row[j] - mean[j]) / stdv[j];
        y = fminf(fmaxf(y, -5.f), 5.f);
        store_opt<T>(out0, ld0, r, j, y);
        store_opt<T>(out1, ld1, r, j, y);
        store_opt<T>(out2, ld2, r, j, y);
    }
}

// 16-byte variant: one f32x4 per thread, 8/16-byte stores
template <typename T>
__global__ __launch_bounds__(256) void rms_normalize4_kernel(const float* __restrict__ src, int64_t ld_src, int D,
                                                             const int32_t* __restrict__ idx, int remap_h, int remap_n,
                                                             int M, const float* __restrict__ mean,
                                                             const float* __restrict__ stdv, void* out0, int64_t ld0,
                                                             void* out1, int64_t ld1, void* out2, int64_t ld2) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r = blockIdx.y * 4 + ty;
    const int j = (blockIdx.x * 64 + tx) * 4;
    if (r >= M || j >= D) return;
    const int64_t p = map_row(r, idx, remap_h, remap_n);
    const f32x4 x = *reinterpret_cast<const f32x4*>(src + p * ld_src + j);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + j), sd = *reinterpret_cast<const f32x4*>(stdv + j);
    f32x4 y;
#pragma unroll
    for (int c = 0; c < 4; ++c) y[c] = fminf(fmaxf((x[c] - mu[c]) / sd[c], -5.f), 5.f);
    void* outs[3] = {out0, out1, out2};
    const int64_t lds[3] = {ld0, ld1, ld2};
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        if (!outs[o]) continue;
        if constexpr (sizeof(T) == 2) {
            typename V16<T>::x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = from_f32<T>(y[c]);
            *reinterpret_cast<typename V16<T>::x4*>(reinterpret_cast<T*>(outs[o]) + (int64_t)r * lds[o] + j) = v;
        } else {
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(outs[o]) + (int64_t)r * lds[o] + j) = y;
        }
    }
}

__global__ void rms_unnormalize_kernel(const double* __restrict__ state, const float* __restrict__ x,
                                       float* __restrict__ y, int64_t n) {
    const float mean = (float)state[0], sd = sqrtf((float)state[1] + 1e-5f);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float c = fminf(fmaxf(x[i], -5.f), 5.f);
        y[i] = sd * c + mean;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, int64_t ld_src, int D,
                                                          const int32_t* __restrict__ idx, int remap_h, int remap_n,
                                                          int M, T* __restrict__ dst, int64_t ld_dst) {
    // a group of W lanes walks one row; W = 64 for wide rows, fewer for narrow ones (set by blockDim.x)
    const int W = blockDim.x, rpb = blockDim.y;
    const int r = blockIdx.x * rpb + threadIdx.y;
    if (r >= M) return;
    const int64_t p = map_row(r, idx, remap_h, remap_n);
    for (int j = threadIdx.x; j < D; j += W) dst[(int64_t)r * ld_dst + j] = from_f32<T>(src[p * ld_src + j]);
}

// fields of one minibatch gathered by ONE launch: desc[f] = {src, ld_src, D, dst, ld_dst, dst_dtype} (int64 each)
// All fields of 16 minibatch rows per workgroup: the 16 row indices are mapped once (LDS), then the (row, column) items of
// every field are spread over the 256 threads - consecutive threads take consecutive columns of a row (narrow fields:
// consecutive rows) - so every load of a thread is independent of the others and nearly all lanes are busy whatever the
// field width (1 ... 64 columns).
__global__ __launch_bounds__(256) void gather_multi_kernel(const int64_t* __restrict__ desc, int n_fields,
                                                           const int32_t* __restrict__ idx, int remap_h, int remap_n, int M) {
    __shared__ int64_t prow[16];
    const int r0 = blockIdx.x * 16;
    if (threadIdx.x < 16 && r0 + threadIdx.x < M) prow[threadIdx.x] = map_row(r0 + threadIdx.x, idx, remap_h, remap_n);
    __syncthreads();
    const int nrows = min(16, M - r0);
    for (int f = 0; f < n_fields; ++f) {
        const int64_t* d = desc + 6 * f;
        const float* src = reinterpret_cast<const float*>(d[0]);
        const int64_t ld_src = d[1], ld_dst = d[4];
        const uint32_t D = (uint32_t)d[2];
        const int dt = (int)d[5];
        const uint32_t items = (uint32_t)nrows * D;
        for (uint32_t i = threadIdx.x; i < items; i += 256) {
            const uint32_t rr = i / D, j = i - rr * D;
            const float v = src[prow[rr] * ld_src + j];
            if (dt == ASE_BF16) reinterpret_cast<bf16_t*>(d[3])[(int64_t)(r0 + rr) * ld_dst + j] = (bf16_t)v;
            else if (dt == ASE_F16) reinterpret_cast<f16_t*>(d[3])[(int64_t)(r0 + rr) * ld_dst + j] = from_f32<f16_t>(v);
            else reinterpret_cast<float*>(d[3])[(int64_t)(r0 + rr) * ld_dst + j] = v;
        }
    }
}

// gather_multi with the identity row map = a multi-tensor f32 -> storage-type conversion (the exact gradient-penalty chain
// handed to the 16-bit launches: 4096 x ~4000 values per step).  8 values per thread - two 16-byte loads, one 16-byte store -
// when the field allows it (width and leading dimensions in whole chunks, aligned bases), else element by element.
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
    const T x = from_f32<T>(a), y = from_f32<T>(b);
    return (uint32_t)__builtin_bit_cast(uint16_t, x) | ((uint32_t)__builtin_bit_cast(uint16_t, y) << 16);
}
__global__ __launch_bounds__(256) void convert_multi_kernel(const int64_t* __restrict__ desc, int M) {
    const int64_t* d = desc + 6 * blockIdx.y;
    const float* src = reinterpret_cast<const float*>(d[0]);
    const int64_t ld_src = d[1], ld_dst = d[4];
    const int D = (int)d[2], dt = (int)d[5];
    char* dst = reinterpret_cast<char*>(d[3]);
    const bool vec = dt != ASE_F32 && D % 8 == 0 && ld_src % 4 == 0 && ld_dst % 8 == 0 && ((uintptr_t)src & 15) == 0 &&
                     ((uintptr_t)dst & 15) == 0;
    const int chunks = vec ? D / 8 : D;
    const int64_t total = (int64_t)M * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / chunks;
        const int c = (int)(i - r * chunks);
        if (vec) {
            const float4 a = *reinterpret_cast<const float4*>(src + r * ld_src + c * 8);
            const float4 b = *reinterpret_cast<const float4*>(src + r * ld_src + c * 8 + 4);
            uint4 o;
            if (dt == ASE_BF16) o = make_uint4(pack2<bf16_t>(a.x, a.y), pack2<bf16_t>(a.z, a.w), pack2<bf16_t>(b.x, b.y), pack2<bf16_t>(b.z, b.w));
            else o = make_uint4(pack2<f16_t>(a.x, a.y), pack2<f16_t>(a.z, a.w), pack2<f16_t>(b.x, b.y), pack2<f16_t>(b.z, b.w));
            *reinterpret_cast<uint4*>(dst + (r * ld_dst + c * 8) * 2) = o;
        } else {
            const float v = src[r * ld_src + c];
            if (dt == ASE_BF16) reinterpret_cast<bf16_t*>(dst)[r * ld_dst + c] = from_f32<bf16_t>(v);
            else if (dt == ASE_F16) reinterpret_cast<f16_t*>(dst)[r * ld_dst + c] = from_f32<f16_t>(v);
            else reinterpret_cast<float*>(dst)[r * ld_dst + c] = v;
        }
    }
}

}  // namespace

extern "C" int ase_hip_gather_multi(const int64_t* desc, int n_fields, const int32_t* idx, int remap_h, int remap_n,
                                    int M, void* stream) {
    ASE_CHECK_ARG(desc && n_fields > 0 && M > 0, "gather_multi: null/empty operand");
    if (idx == nullptr && remap_h <= 0) {                       // identity row map: a plain conversion, vectorised
        ASE_LAUNCH(convert_multi_kernel, dim3(1024, n_fields), dim3(256), 0, (hipStream_t)stream, desc, M);
        ASE_CHECK_LAUNCH("gather_multi");
        return ASE_OK;
    }
    ASE_LAUNCH(gather_multi_kernel, dim3((M + 15) / 16), dim3(256), 0, (hipStream_t)stream, desc, n_fields, idx, remap_h,
               remap_n, M);
    ASE_CHECK_LAUNCH("gather_multi");
    return ASE_OK;
}

extern "C" int ase_hip_rms_moments(const float* src, int64_t ld_src, int D, const int32_t* idx, int remap_h,
                                   int remap_n, int M, const double* state, double* sums, void* stream) {
    ASE_CHECK_ARG(src && state && sums && D > 0 && M > 0, "rms_moments: null/empty operand");
    // (the 16-byte variant measured SLOWER here: 18.1 vs 16.2 us on 4096 x 1400 - a quarter of the workgroups for the
    //  same dependent index -> row loads; kept for wide, un-gathered inputs)
    if (idx == nullptr && D % 4 == 0 && ld_src % 4 == 0 && ((uintptr_t)src % 16) == 0) {
        const dim3 grid((D / 4 + 63) / 64, (M + kRowsPerBlock - 1) / kRowsPerBlock);
        ASE_LAUNCH(rms_moments4_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, D, idx, remap_h,
                           remap_n, M, state, sums);
    } else {
        RmsStreams S = {};
        S.src[0] = src; S.ld[0] = ld_src; S.idx[0] = idx; S.rh[0] = remap_h; S.rn[0] = remap_n; S.sums[0] = sums;
        const dim3 grid((D + 63) / 64, (M + kRowsPerBlock - 1) / kRowsPerBlock, 1);
        ASE_LAUNCH(rms_moments_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, S, D, M, state);
    }
    ASE_CHECK_LAUNCH("rms_moments");
    return ASE_OK;
}

extern "C" int ase_hip_rms_moments_multi(const float* const* srcs, const int64_t* ld_srcs, const int32_t* const* idxs,
                                         const int* remap_h, const int* remap_n, double* const* sums, int n_streams, int D,
                                         int M, const double* state, void* stream) {
    ASE_CHECK_ARG(srcs && ld_srcs && idxs && remap_h && remap_n && sums && state && n_streams >= 1 && n_streams <= 4 && D > 0 && M > 0,
                  "rms_moments_multi: null/empty operand (1..4 streams)");
    RmsStreams S = {};
    for (int s = 0; s < n_streams; ++s) {
        ASE_CHECK_ARG(srcs[s] && sums[s], "rms_moments_multi: stream %d: null operand", s);
        S.src[s] = srcs[s]; S.ld[s] = ld_srcs[s]; S.idx[s] = idxs[s]; S.rh[s] = remap_h[s]; S.rn[s] = remap_n[s]; S.sums[s] = sums[s];
    }
    const dim3 grid((D + 63) / 64, (M + kRowsPerBlock - 1) / kRowsPerBlock, n_streams);
    ASE_LAUNCH(rms_moments_multi_kernel, grid, dim3(256), 0, (hipStream_t)stream, S, D, M, state);
    ASE_CHECK_LAUNCH("rms_moments_multi");
    return ASE_OK;
}

extern "C" int ase_hip_rms_normalize_multi(const float* const* srcs, const int64_t* ld_srcs, const int32_t* const* idxs,
                                           const int* remap_h, const int* remap_n, const float* const* means,
                                           const float* const* stds, void* const* outs, const int64_t* ld_outs, int n_streams,
                                           int D, int M, int dtype, void* stream) {
    ASE_CHECK_ARG(srcs && ld_srcs && idxs && remap_h && remap_n && means && stds && outs && ld_outs && n_streams >= 1 &&
                      n_streams <= 4 && D > 0 && M > 0, "rms_normalize_multi: null/empty operand (1..4 streams)");
    ASE_CHECK_ARG(dtype == ASE_BF16 || dtype == ASE_F32 || dtype == ASE_F16, "rms_normalize_multi: bad dtype %d", dtype);
    const int es = ase_elem_size(dtype);
    RmsStreams S = {};
    for (int s = 0; s < n_streams; ++s) {
        ASE_CHECK_ARG(srcs[s] && means[s] && stds[s] && outs[s], "rms_normalize_multi: stream %d: null operand", s);
        ASE_CHECK_ARG(D % 4 == 0 && ld_srcs[s] % 4 == 0 && ((uintptr_t)srcs[s] % 16) == 0 && ((uintptr_t)means[s] % 16) == 0 &&
                          ((uintptr_t)stds[s] % 16) == 0 && ld_outs[s] % 4 == 0 && ((uintptr_t)outs[s] % (4 * es)) == 0,
                      "rms_normalize_multi: stream %d needs 16-byte rows (D %% 4 == 0, aligned pointers / pitches)", s);
        S.src[s] = srcs[s]; S.ld[s] = ld_srcs[s]; S.idx[s] = idxs[s]; S.rh[s] = remap_h[s]; S.rn[s] = remap_n[s];
        S.mean[s] = means[s]; S.stdv[s] = stds[s]; S.out[s] = outs[s]; S.ld_out[s] = ld_outs[s];
    }
    const dim3 grid((D / 4 + 63) / 64, (M + 3) / 4, n_streams);
    ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH(rms_normalize_multi_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, S, D, M);
        return ASE_OK;
    });
    ASE_CHECK_LAUNCH("rms_normalize_multi");
    return ASE_OK;
}

extern "C" int ase_hip_rms_finalize(double* state, int D, const double* sums, const int32_t* counts, int n_streams,
                                    float* mean_out, float* std_out, void* stream) {
    // `counts` is a HOST array of n_streams global row counts; all streams of one call share one count.
    ASE_CHECK_ARG(state && mean_out && std_out && D > 0 && n_streams >= 0, "rms_finalize: null/empty operand");
    int count = 0;
    if (n_streams > 0) {
        ASE_CHECK_ARG(sums && counts, "rms_finalize: sums/counts missing");
        count = counts[0];
        for (int s = 1; s < n_streams; ++s)
            ASE_CHECK_ARG(counts[s] == count, "rms_finalize: all streams of one call must have the same row count");
        ASE_CHECK_ARG(count > 1, "rms_finalize: need at least 2 rows for an unbiased variance");
    }
    // one workgroup (the count is read before it is rewritten); 1024 threads: the 1400 AMP columns take 2 passes, not 6
    ASE_LAUNCH(rms_finalize_kernel, dim3(1), dim3(D > 256 ? 1024 : 256), 0, (hipStream_t)stream, state, D, sums, count,
                       n_streams, mean_out, std_out);
    ASE_CHECK_LAUNCH("rms_finalize");
    return ASE_OK;
}

extern "C" int ase_hip_rms_normalize(const float* src, int64_t ld_src, int D, const int32_t* idx, int remap_h,
                                     int remap_n, int M, const float* mean, const float* stdv, void* out0,
                                     int64_t ld0, void* out1, int64_t ld1, void* out2, int64_t ld2, int dtype,
                                     void* stream) {
    ASE_CHECK_ARG(src && mean && stdv && out0 && D > 0 && M > 0, "rms_normalize: null/empty operand");
    const int es = ase_elem_size(dtype);
    auto ok = [&](void* o, int64_t ld) { return o == nullptr || (ld % 4 == 0 && ((uintptr_t)o % (4 * es)) == 0); };
    const bool wide = D % 4 == 0 && ld_src % 4 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)mean % 16) == 0 &&
                      ((uintptr_t)stdv % 16) == 0 && ok(out0, ld0) && ok(out1, ld1) && ok(out2, ld2);
    const dim3 g4((D / 4 + 63) / 64, (M + 3) / 4), grid((D + 255) / 256, (M + 3) / 4);
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        if (wide)
            ASE_LAUNCH(rms_normalize4_kernel<T>, g4, dim3(256), 0, (hipStream_t)stream, src, ld_src, D, idx, remap_h, remap_n, M, mean,
                       stdv, out0, ld0, out1, ld1, out2, ld2);
        else
            ASE_LAUNCH(rms_normalize_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, src, ld_src, D, idx, remap_h, remap_n, M, mean,
                       stdv, out0, ld0, out1, ld1, out2, ld2);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "rms_normalize: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("rms_normalize");
    return ASE_OK;
}

extern "C" int ase_hip_rms_unnormalize(const double* state, const float* x, float* y, int64_t n, void* stream) {
    ASE_CHECK_ARG(state && x && y && n > 0, "rms_unnormalize: null/empty operand");
    const int blocks = (int)min((int64_t)2048, (n + 255) / 256);
    ASE_LAUNCH(rms_unnormalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, state, x, y, n);
    ASE_CHECK_LAUNCH("rms_unnormalize");
    return ASE_OK;
}

extern "C" int ase_hip_gather_rows(const float* src, int64_t ld_src, int D, const int32_t* idx, int remap_h,
                                   int remap_n, int M, void* dst, int64_t ld_dst, int dst_dtype, void* stream) {
    ASE_CHECK_ARG(src && dst && D > 0 && M > 0, "gather_rows: null/empty operand");
    int W = 64;
    while (W > 1 && W / 2 >= D) W /= 2;
    const dim3 block(W, 256 / W);
    const dim3 grid((M + block.y - 1) / block.y);
    const int rc = ase_dispatch_storage(dst_dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH(gather_rows_kernel<T>, grid, block, 0, (hipStream_t)stream, src, ld_src, D, idx, remap_h, remap_n, M, (T*)dst, ld_dst);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "gather_rows: bad dtype %d", dst_dtype);
    ASE_CHECK_LAUNCH("gather_rows");
    return ASE_OK;
}
