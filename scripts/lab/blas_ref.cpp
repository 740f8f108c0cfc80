// Vendor-library reference for the NT shapes of the update: rocBLAS gemm_ex (bf16 in / out, f32 accumulate) on the same
// row-major operands C[M, N] = A[M, K] B[N, K]^T, timed with HIP events.  Tuning aid only (scripts/lab): the product does
// not link rocBLAS; this tells how far the hand-written tiles are from what the vendor's tuned kernels reach on this chip.
//   build: make -C scripts/lab blas_ref        run: scripts/lab/blas_ref M N K [reps] [tn]
//   tn: the weight-gradient shape G[N, K] (f32) = A[M, N]^T B[M, K] (contraction over the M rows), bf16 in / f32 out
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
__global__ void fill_kernel(__bf16* x, int64_t n, uint64_t seed, float scale) {      // the fill of gemm_lab.cpp: uniform [-scale, scale)
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        x[i] = (__bf16)(((float)(z >> 40) * (1.0f / 16777216.0f) * 2.f - 1.f) * scale);
    }
}
#define RB(x) do { rocblas_status s_ = (x); if (s_ != rocblas_status_success) { printf("rocBLAS error %d at %s:%d\n", (int)s_, __FILE__, __LINE__); exit(3); } } while (0)

int main(int argc, char** argv) {
    if (argc < 4) { printf("usage: blas_ref M N K [reps]\n"); return 1; }
    const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]), reps = argc > 4 ? atoi(argv[4]) : 20;
    const bool tn = argc > 5 && !strcmp(argv[5], "tn");
    hipStream_t st; CK(hipStreamCreate(&st));
    rocblas_handle h; RB(rocblas_create_handle(&h)); RB(rocblas_set_stream(h, st));
    void *A, *B, *C;
    if (tn) {
        // row-major A[M, N], B[M, K] -> G[N, K]; column-major view: G'[K, N] = B'[K, M] A'[N, M]^T
        void *G;
        CK(hipMalloc(&A, (size_t)M * N * 2)); CK(hipMalloc(&B, (size_t)M * K * 2)); CK(hipMalloc(&G, (size_t)N * K * 4));
        fill_kernel<<<1024, 256, 0, st>>>((__bf16*)A, (int64_t)M * N, 1, 1.0f);
        fill_kernel<<<1024, 256, 0, st>>>((__bf16*)B, (int64_t)M * K, 2, 0.05f);
        const float alpha = 1.f, beta = 0.f;
        auto run = [&]() {
            RB(rocblas_gemm_ex(h, rocblas_operation_none, rocblas_operation_transpose, K, N, M, &alpha, B, rocblas_datatype_bf16_r, K,
                               A, rocblas_datatype_bf16_r, N, &beta, G, rocblas_datatype_f32_r, K, G, rocblas_datatype_f32_r, K,
                               rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0));
        };
        for (int i = 0; i < 3; ++i) run();
        CK(hipStreamSynchronize(st));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("rocBLAS TN %6d x %5d x %5d (f32 out): %8.1f us %8.1f TF/s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
        return 0;
    }
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)M * N * 2));
    if (getenv("LAB_CONST")) {          // constant operands: far fewer toggling bits => higher sustained clocks; NOT comparable
        CK(hipMemset(A, 0x3c, (size_t)M * K * 2)); CK(hipMemset(B, 0x3c, (size_t)N * K * 2));
    } else {
        fill_kernel<<<1024, 256, 0, st>>>((__bf16*)A, (int64_t)M * K, 1, 1.0f);
        fill_kernel<<<1024, 256, 0, st>>>((__bf16*)B, (int64_t)N * K, 2, 0.05f);
    }
    const float alpha = 1.f, beta = 0.f;
    // column-major view: C'[N, M] = B'[K, N]^T A'[K, M]
    auto run = [&]() {
        RB(rocblas_gemm_ex(h, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &alpha, B, rocblas_datatype_bf16_r, K,
                           A, rocblas_datatype_bf16_r, K, &beta, C, rocblas_datatype_bf16_r, N, C, rocblas_datatype_bf16_r, N,
                           rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0));
    };
    for (int i = 0; i < 3; ++i) run();
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) run();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("rocBLAS NT %6d x %5d x %5d: %8.1f us %8.1f TF/s\n", M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
    return 0;
}
