// Shared device/host helpers for libase_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ase_hip.h"
#include "prog.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16_t;                                    // IEEE half: 10 explicit mantissa bits (bf16: 7), same MFMA rate
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

void ase_set_error(const char* fmt, ...);

#define ASE_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ase_set_error(__VA_ARGS__);          \
            return ASE_EINVAL;                   \
        }                                        \
    } while (0)

#define ASE_CHECK_LAUNCH(name)                                              \
    do {                                                                    \
        hipError_t e__ = hipGetLastError();                                 \
        if (e__ != hipSuccess) {                                            \
            ase_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return ASE_ELAUNCH;                                             \
        }                                                                   \
    } while (0)

// f32 storage whose matrix products run as THREE bf16 MFMAs on a hi/lo split (a = hi + lo, hi = bf16(a),
// lo = bf16(a - hi); a*b ~= hi*hi + hi*lo + lo*hi): ~16 mantissa bits per operand at 1/3 of the bf16 MFMA rate
// instead of the 1/16 of the exact-f32 MFMA (gfx950 has no TF32).
struct f32s_t { float v; };
// f32 storage whose matrix products run as THREE f16 MFMAs on a hi/lo split of SCALED operands: sx = x * 2^e (exact),
// hi = half(sx), lo = half(sx - hi); sx*sy ~= hi*hi + hi*lo + lo*hi, undone by 2^-(ea + eb) in the epilogue's alpha.  Half
// keeps 11 significant bits per part (bf16: 8), so hi + lo carries ~22 bits against the bf16 split's ~16 at the same three
// MFMAs - but half's exponent range is narrow: the caller chooses ea / eb so that the operands' magnitudes land in
// [2^-2, 2^15] (above: the conversion saturates; below: lo turns subnormal and the product degrades towards 2^-13).
struct f32h_t { float v; };

// ---- storage type conversion --------------------------------------------------------------
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
__device__ __forceinline__ float to_f32(f32s_t x) { return x.v; }
__device__ __forceinline__ float to_f32(f32h_t x) { return x.v; }
__device__ __forceinline__ float to_f32(f16_t x) { return (float)x; }

template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }  // RNE
template <> __device__ __forceinline__ f32s_t from_f32<f32s_t>(float x) { return f32s_t{x}; }
template <> __device__ __forceinline__ f32h_t from_f32<f32h_t>(float x) { return f32h_t{x}; }
// f16: RNE, saturating at the largest finite half (a scaled gradient that overflows must not turn into inf -> NaN)
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float x) { return (f16_t)__builtin_amdgcn_fmed3f(x, -65504.f, 65504.f); }

// ---- the two 16-bit storage types share every kernel: vector types and the matrix instruction by type -----------
template <typename T> struct V16;
template <> struct V16<bf16_t> { typedef bf16x8 x8; typedef bf16x4 x4; };
template <> struct V16<f16_t> { typedef f16x8 x8; typedef f16x4 x4; };
// (instantiated for 4-byte storage types only in dead branches)
template <> struct V16<float> { typedef bf16x8 x8; typedef bf16x4 x4; };
template <typename T> __device__ __forceinline__ f32x16 mfma16(typename V16<T>::x8 a, typename V16<T>::x8 b, f32x16 c);
template <> __device__ __forceinline__ f32x16 mfma16<bf16_t>(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x16 mfma16<f16_t>(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// ---- host-side dispatch over the storage types of activations / shadow weights ----------------------------------
template <typename T> struct StorageTag { typedef T type; };
template <typename F> inline int ase_dispatch_storage(int dtype, F&& f) {
    switch (dtype) {
        case ASE_F32: return f(StorageTag<float>{});
        case ASE_BF16: return f(StorageTag<bf16_t>{});
        case ASE_F16: return f(StorageTag<f16_t>{});
        default: return ASE_EUNSUPPORTED;
    }
}
inline int ase_elem_size(int dtype) { return (dtype == ASE_BF16 || dtype == ASE_F16) ? 2 : 4; }

// ---- dataset row map (see ase_hip.h) -------------------------------------------------------
__device__ __forceinline__ int64_t map_row(int r, const int32_t* __restrict__ idx, int remap_h, int remap_n) {
    // 32-bit unsigned arithmetic (row numbers are int32): a 64-bit signed division per row cost more than the row's loads
    uint32_t p = idx ? (uint32_t)idx[r] : (uint32_t)r;
    if (remap_h > 0) {
        const uint32_t env = p / (uint32_t)remap_h;
        const uint32_t t = p - env * (uint32_t)remap_h;
        p = t * (uint32_t)remap_n + env;
    }
    return (int64_t)p;
}

// ---- reductions ------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum within aligned groups of W lanes (W power of two <= 64)
template <int W> __device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum of NV doubles; result valid in thread 0. blockDim.x <= 1024.
template <int NV> __device__ __forceinline__ void block_sum(double (&v)[NV], double* smem /* [NV*16] */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_sum(v[i]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) smem[i * 16 + wid] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0;
            for (int w = 0; w < nw; ++w) s += smem[i * 16 + w];
            v[i] = s;
        }
    }
}

__device__ __forceinline__ void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ---- scale records of the dynamic loss scale (ase_hip.h): {factor, overflow count} ---------------------------------------------------
// A launch that was given a record multiplies `factor` into its scale and reports what it STORED: an element that is non-finite or sits at
// the storage type's saturation value (the f16 conversions above saturate at +-65504 where autocast would produce inf; bf16 goes to
// inf; f32 storage: NaN / inf).  |x| of a 16-bit pattern orders like an unsigned integer.
template <typename T> __device__ __forceinline__ constexpr uint32_t ovf_threshold() { return 0x7F80u; }      // the first non-finite bf16
template <> __device__ __forceinline__ constexpr uint32_t ovf_threshold<f16_t>() { return 0x7BFFu; }          // |65504| in IEEE half
__device__ __forceinline__ bool ovf_hit1(f16_t x) { return (uint32_t)(__builtin_bit_cast(unsigned short, x) & 0x7FFFu) >= 0x7BFFu; }
__device__ __forceinline__ bool ovf_hit1(bf16_t x) { return (uint32_t)(__builtin_bit_cast(unsigned short, x) & 0x7FFFu) >= 0x7F80u; }
__device__ __forceinline__ bool ovf_hit1(float x) { return !(__builtin_fabsf(x) <= 3.402823466e38f); }
// one atomic per wave that saw one, from its first active lane (the count's value is unspecified beyond zero / non-zero)
__device__ __forceinline__ void ovf_report(float* rec, bool bad) {
    if (rec && __builtin_amdgcn_ballot_w64(bad) != 0 &&
        (int)(threadIdx.x & 63) == __builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
        atomic_add_f32(rec + 1, 1.f);
}
