// C entry points of the NT matrix-core kernels (gfx950 / CDNA4): argument checks, kernel choice, debug hooks.  The kernels
// live in gemm_nt_kernels.h and are instantiated per storage type in gemm_nt_<type>.hip; the weight-gradient kernels and the
// shadow refresh in gemm_tn.hip.
#include "gemm_nt.h"
#include <math.h>

namespace ase_nt {

unsigned long long* g_nt_prof = nullptr;
int g_nt_prof_clk = 0;

int nt_choice(int M, int N, int K, int es, bool b16) {
    if (N <= 64) return 0;
    const int t256 = ((M + 255) / 256) * ((N + 255) / 256);
    const bool big = N % 256 == 0 && t256 >= 192 && (t256 <= 256 || t256 % 256 == 0 || t256 >= 1024);
    if (big) {
        if (!(b16 && (K * es) % 128 == 0)) return 3;
        // one round that leaves CUs idle (12288 x 1024: 192 tiles): 192-row tiles of the same schedule fill them
        const int t192 = ((M + 191) / 192) * (N / 256);
        return (t256 < 256 && t192 <= 256 && t192 > t256) ? 6 : 2;
    }
    if ((K * es) % 128 != 0) return 1;
    // grids that do not fill the phased kernel's rounds: the LARGEST of the 128 x 128 / 64 x 128 / 64 x 64 tiles that still
    // gives two workgroups per CU (measured: 4096 x 1024 x 1024  21.1 -> 17.6 us, 4096 x 512 x 1024  19.3 -> 11.4 us,
    // 2048 x 1024 x 1024 - one rank's shard at 8 GPUs -  19.3 -> 11.4 us; 16384 x 512 stays on 128 x 128)
    const int t128 = ((M + 127) / 128) * ((N + 127) / 128), t64x128 = ((M + 63) / 64) * ((N + 127) / 128);
    if (t128 >= 512) return 1;
    if (t64x128 >= 512) return 4;
    return 5;
}

}  // namespace ase_nt

using namespace ase_nt;

extern "C" int ase_hip_debug_nt_profile(void* buf) {
    g_nt_prof = reinterpret_cast<unsigned long long*>(buf);
    return ASE_OK;
}

extern "C" int ase_hip_debug_nt_profile_clock(int shader_clock) {
    g_nt_prof_clk = shader_clock ? 1 : 0;
    return ASE_OK;
}

extern "C" int ase_hip_gemm_nt_kernel_id(int M, int N, int K, int dtype) {
    dtype &= 0xFF;
    return nt_choice(M, N, K, ase_elem_size(dtype), ase_elem_size(dtype) == 2);
}

extern "C" int ase_hip_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                               const float* bias, const void* aux, int64_t ldaux, int aux_split, int aux_delta,
                               float* colsum, int colsum_n, void* mask_out, int64_t ldmask, int M, int N, int K, int act,
                               int aux_mode, int out_f32, float alpha, float* alpha_dev, int dtype_word, void* stream) {
    const int dtype = dtype_word & 0xFF, ea = (dtype_word >> 8) & 0xFF, eb = (dtype_word >> 16) & 0xFF;
    const int es = ase_elem_size(dtype);
    ASE_CHECK_ARG(dtype == ASE_F32 || dtype == ASE_BF16 || dtype == ASE_F32X3 || dtype == ASE_F16 || dtype == ASE_F32H3,
                  "gemm_nt: bad dtype %d", dtype);
    ASE_CHECK_ARG(ea >= 0 && ea <= 24 && eb >= 0 && eb <= 24 && (dtype == ASE_F32H3 || (ea | eb) == 0),
                  "gemm_nt: operand scale exponents (bits 8-15 / 16-23 of dtype) are 0..24 and belong to ASE_F32H3 only");
    ASE_CHECK_ARG(A && B && C && M > 0 && N > 0 && K > 0, "gemm_nt: null/empty operand (M=%d N=%d K=%d)", M, N, K);
    ASE_CHECK_ARG((K * es) % 64 == 0, "gemm_nt: K=%d is not a multiple of %d elements", K, 64 / es);
    ASE_CHECK_ARG(lda >= K && ldb >= K && ldc >= N, "gemm_nt: leading dimension too small");
    ASE_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && (lda * es) % 16 == 0 && (ldb * es) % 16 == 0,
                  "gemm_nt: A/B must be 16-byte aligned with 16-byte row pitch");
    ASE_CHECK_ARG(aux_mode == ASE_AUX_NONE || aux != nullptr, "gemm_nt: aux_mode %d without aux", aux_mode);
    const bool bits = aux_mode == ASE_AUX_RELU_BITS;
    ASE_CHECK_ARG(N % 4 == 0 && ((uintptr_t)C % 16) == 0 && (ldc * (out_f32 ? 4 : es)) % 8 == 0 &&
                      (aux == nullptr || bits || (((uintptr_t)aux % 8) == 0 && (ldaux * es) % 8 == 0)),
                  "gemm_nt: C / aux must allow 8/16-byte row-vector access (N %% 4 == 0, aligned pitches)");
    ASE_CHECK_ARG(!bits || (((uintptr_t)aux % 4) == 0 && ldaux * 32 >= N), "gemm_nt: bit mask needs ldaux >= N / 32 words");
    const bool smooth = act >= ASE_ACT_SILU;
    ASE_CHECK_ARG(act >= ASE_ACT_NONE && act <= ASE_ACT_SOFTPLUS, "gemm_nt: unknown activation %d", act);
    ASE_CHECK_ARG((aux_mode & 0xFF) <= ASE_AUX_PREACT && ((aux_mode & 0xFF) == ASE_AUX_PREACT || (aux_mode >> 8) == 0) &&
                      (aux_mode >> 8) <= ASE_ACT_SOFTPLUS, "gemm_nt: bad aux_mode 0x%x", aux_mode);
    ASE_CHECK_ARG(mask_out == nullptr || smooth || (N % 32 == 0 && ((uintptr_t)mask_out % 4) == 0 && ldmask * 32 >= N),
                  "gemm_nt: mask_out needs N %% 32 == 0 and ldmask >= N / 32 words");
    ASE_CHECK_ARG(mask_out == nullptr || !smooth || (ldmask >= N && ((uintptr_t)mask_out % 16) == 0 && (ldmask * es) % 8 == 0),
                  "gemm_nt: the pre-activation twin needs ldmask >= N elements, 16-byte alignment");
    NTParams p;
    p.A = (const char*)A; p.lda = lda * es;
    p.B = (const char*)B; p.ldb = ldb * es;
    p.C = (char*)C; p.ldc = ldc * (out_f32 ? 4 : es);
    p.mask_out = smooth ? nullptr : (uint32_t*)mask_out; p.ldmask = ldmask;
    p.pre_out = smooth ? (char*)mask_out : nullptr; p.ldpre = ldmask * es;
    p.bias = bias; p.aux = (const char*)aux; p.ldaux = bits ? ldaux * 4 : ldaux * es; p.aux_split = aux_split > 0 ? aux_split : M; p.aux_delta = aux_delta; p.colsum = colsum; p.colsum_n = colsum ? colsum_n : 0;
    p.M = M; p.N = N; p.K = K; p.act = act; p.aux_mode = aux_mode; p.out_f32 = out_f32; p.alpha = alpha; p.alpha_dev = alpha_dev;
    p.tiles_m = p.tiles_n = 0;
    p.prof = nullptr;
    p.prof_clk = 0;
    p.sa = 1.f;
    if (dtype == ASE_BF16) return dispatch_nt_bf16(p, (hipStream_t)stream);
    if (dtype == ASE_F16) return dispatch_nt_f16(p, (hipStream_t)stream);
    if (dtype == ASE_F32X3) return dispatch_nt_x3(p, (hipStream_t)stream);
    if (dtype == ASE_F32H3) {
        p.sa = ldexpf(1.f, ea);            // (B: pre-split and scaled by 2^eb by ase_hip_refresh_shadow)
        p.alpha = ldexpf(alpha, -(ea + eb));          // exact: powers of two
        return dispatch_nt_h3(p, (hipStream_t)stream);
    }
    return dispatch_nt_f32(p, (hipStream_t)stream);
}

