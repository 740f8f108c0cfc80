// Stand-alone driver for the matrix-core kernels of libase_hip.so (no Python / torch: starts in milliseconds on the GPU
// box).  Checks sampled rows against a naive f32 device reference and times repeated launches with HIP events.
//   build: make -C scripts/lab        run: scripts/lab/gemm_lab nt|tn M N K [reps] [aux] [relu]
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <cmath>
#include <algorithm>
#include "../../include/ase_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef __bf16 bf16_t;

__global__ void fill_kernel(bf16_t* x, int64_t n, uint64_t seed, float scale) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const float u = (float)(z >> 40) * (1.0f / 16777216.0f) * 2.f - 1.f;   // uniform [-1, 1)
        x[i] = (bf16_t)(u * scale);
    }
}
__global__ void fillf_kernel(float* x, int64_t n, uint64_t seed, float scale) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z ^= z >> 31;
    x[i] = ((float)(z >> 40) * (1.0f / 16777216.0f) * 2.f - 1.f) * scale;
}

__global__ void pack_bits_kernel(const bf16_t* aux, uint32_t* bits, int M, int N) {
    const int64_t words = (int64_t)M * ((N + 31) / 32);
    for (int64_t w = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; w < words; w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = w / ((N + 31) / 32); const int j = (int)(w % ((N + 31) / 32));
        uint32_t v = 0;
        for (int b = 0; b < 32 && j * 32 + b < N; ++b) v |= ((float)aux[m * N + j * 32 + b] > 0.f ? 1u : 0u) << b;
        bits[w] = v;
    }
}

// NT reference on sampled rows: out[s, n] = mask(act(sum_k A[row_s, k] B[n, k] + bias[n]))
__global__ void ref_nt_kernel(const bf16_t* A, const bf16_t* B, const float* bias, const bf16_t* aux, const int* rows,
                              float* out, int N, int K, int relu) {
    const int s = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int m = rows[s];
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += (float)A[(int64_t)m * K + k] * (float)B[(int64_t)n * K + k];
    acc += bias[n];
    if (relu) acc = fmaxf(acc, 0.f);
    if (aux) acc = ((float)aux[(int64_t)m * N + n] > 0.f) ? acc : 0.f;
    out[(int64_t)s * N + n] = acc;
}
// TN reference on sampled output rows n: G[n, k] = sum_m A[m, n] B[m, k]
__global__ void ref_tn_kernel(const bf16_t* A, const bf16_t* B, const int* rows, float* out, int M, int N, int K) {
    const int s = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int n = rows[s];
    float acc = 0.f;
    for (int m = 0; m < M; ++m) acc += (float)A[(int64_t)m * N + n] * (float)B[(int64_t)m * K + k];
    out[(int64_t)s * K + k] = acc;
}

static void print_profile(unsigned long long* prof, int tiles) {
    std::vector<unsigned long long> h(tiles * 4);
    CK(hipMemcpy(h.data(), prof, tiles * 32, hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull, tmax = 0; double s[3] = {0, 0, 0}, mx[3] = {0, 0, 0}; double start_spread = 0; int live = 0;
    for (int i = 0; i < tiles; ++i) { if (!h[i * 4]) continue; ++live; if (h[i * 4] < tmin) tmin = h[i * 4]; if (h[i * 4 + 3] > tmax) tmax = h[i * 4 + 3]; }
    for (int i = 0; i < tiles; ++i) {
        if (!h[i * 4]) continue;
        for (int q = 0; q < 3; ++q) { const double d = (double)(h[i * 4 + q + 1] - h[i * 4 + q]) * 0.01; s[q] += d; if (d > mx[q]) mx[q] = d; }
        const double d0 = (double)(h[i * 4] - tmin) * 0.01; if (d0 > start_spread) start_spread = d0;
    }
    printf("   profile (%d workgroups, us): prologue avg %.2f max %.2f | main loop avg %.2f max %.2f | epilogue+drain avg %.2f max %.2f | "
           "entry spread %.2f | first entry -> last retire %.2f\n", live, s[0] / live, mx[0], s[1] / live, mx[1], s[2] / live, mx[2],
           start_spread, (double)(tmax - tmin) * 0.01);
}

static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- grouped weight gradients: the dense layers of one ASE optimisation step (config 2), one launch vs one per layer
static int run_grouped(int reps) {
    struct Shape { int M, N, K; };
    const Shape shapes[] = {{32768, 1024, 320}, {32768, 1024, 1024}, {32768, 512, 1024},      // actor (2 x 16384 rows)
                            {16384, 1024, 320}, {16384, 1024, 1024}, {16384, 512, 1024},      // critic
                            {16384, 1024, 1408}, {16384, 1024, 1024}, {16384, 512, 1024},     // discriminator (4 x 4096 rows)
                            {32768, 512, 64}, {32768, 256, 512}, {32768, 64, 256},            // style MLP
                            {32768, 32, 512}, {16384, 8, 512}, {16384, 72, 512}};             // heads (padded widths)
    // LAB_TNG_N: how many of the shapes take part (9 = the wide layers only, 11 = + the two mid-size style-MLP layers, 15 = all)
    const int P = getenv("LAB_TNG_N") ? atoi(getenv("LAB_TNG_N")) : (int)(sizeof(shapes) / sizeof(shapes[0])), S = 24;
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<bf16_t*> A(P), B(P); std::vector<float*> G(P), gb(P);
    std::vector<int64_t> prob(16 * P, 0);
    double flops = 0;
    for (int i = 0; i < P; ++i) {
        const Shape& h = shapes[i];
        // LAB_TNG_ALIAS=1: every problem reads the SAME two operand buffers (the largest shapes': 64 + 92 MB, inside the 256 MB
        // Infinity Cache) - the launch's time with its operand traffic served on-die, i.e. how much of it is memory time
        static bf16_t *A0 = nullptr, *B0 = nullptr;
        if (getenv("LAB_TNG_ALIAS") && atoi(getenv("LAB_TNG_ALIAS"))) {
            if (!A0) { CK(hipMalloc(&A0, (int64_t)32768 * 1024 * 2)); CK(hipMalloc(&B0, (int64_t)32768 * 1408 * 2)); }
            A[i] = A0; B[i] = B0;
        } else {
            CK(hipMalloc(&A[i], (int64_t)h.M * h.N * 2)); CK(hipMalloc(&B[i], (int64_t)h.M * h.K * 2));
        }
        CK(hipMalloc(&G[i], (int64_t)h.N * h.K * 4)); CK(hipMalloc(&gb[i], h.N * 4));
        fill_kernel<<<1024, 256, 0, st>>>(A[i], (int64_t)h.M * h.N, 11 + i, 1.0f);
        fill_kernel<<<1024, 256, 0, st>>>(B[i], (int64_t)h.M * h.K, 77 + i, 0.05f);
        CK(hipMemsetAsync(G[i], 0, (int64_t)h.N * h.K * 4, st)); CK(hipMemsetAsync(gb[i], 0, h.N * 4, st));
        int64_t* d = &prob[16 * i];
        d[0] = (int64_t)A[i]; d[1] = h.N; d[2] = (int64_t)B[i]; d[3] = h.K; d[4] = (int64_t)G[i]; d[5] = (int64_t)gb[i];
        d[6] = (i >= 6 && i < 9) ? 12288 : 0; d[7] = h.M; d[8] = h.N; d[9] = h.K; d[10] = h.N; d[11] = h.K; d[12] = h.K; d[13] = h.K;
        const float one = 1.0f; int bits; memcpy(&bits, &one, 4); d[14] = bits;
        flops += 2.0 * h.M * h.N * h.K;
    }
    std::vector<int32_t> work(4 * 4096), red(4 * 4096); int nw = 0, nr = 0;
    if (ase_hip_gemm_tn_grouped_plan(prob.data(), P, 0, work.data(), 4096, &nw, red.data(), 4096, &nr)) { printf("plan failed: %s\n", ase_hip_last_error()); return 3; }
    int mn = 1 << 30, mx = 0; for (int i = 0; i < nw; ++i) { mn = std::min(mn, work[4 * i + 3]); mx = std::max(mx, work[4 * i + 3]); }
    printf("grouped plan: %d problems -> %d work items, K-tiles per item %d..%d\n", P, nw, mn, mx);
    int64_t* dprob; int32_t *dwork, *dred; float* ws = nullptr;
    CK(hipMalloc(&dprob, prob.size() * 8)); CK(hipMalloc(&dwork, (size_t)nw * 16)); CK(hipMalloc(&dred, (size_t)nr * 16));
    CK(hipMemcpy(dprob, prob.data(), prob.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dwork, work.data(), (size_t)nw * 16, hipMemcpyHostToDevice));
    CK(hipMemcpy(dred, red.data(), (size_t)nr * 16, hipMemcpyHostToDevice));
    // LAB_TNG_WS=0: f32 atomics into G; default: partial tiles in a workspace + the reduce kernel
    if (!getenv("LAB_TNG_WS") || atoi(getenv("LAB_TNG_WS"))) CK(hipMalloc(&ws, (size_t)nw * ASE_TN_SLAB * 4));
    printf("reduce entries %d, workspace %s\n", nr, ws ? "on" : "off (atomics)");
    auto grouped = [&]() { if (ase_hip_gemm_tn_grouped(dprob, dwork, nw, dred, nr, ws, nullptr, ASE_BF16, st)) { printf("grouped failed: %s\n", ase_hip_last_error()); exit(3); } };
    auto single = [&]() {
        for (int i = 0; i < P; ++i) {
            const Shape& h = shapes[i];
            if (ase_hip_gemm_tn(A[i], h.N, B[i], h.K, G[i], gb[i], (int)prob[16 * i + 6] == h.M ? 0 : (int)prob[16 * i + 6], h.M, h.N, h.K, h.N, h.K, h.K, h.K, 1.0f, nullptr, ASE_BF16, st)) { printf("tn failed: %s\n", ase_hip_last_error()); exit(3); }
        }
    };
    grouped();
    // verify every problem on sampled output rows + the bias gradient of sampled columns
    int bad = 0;
    for (int i = 0; i < P; ++i) {
        const Shape& h = shapes[i];
        std::vector<int> rows(S); for (int s = 0; s < S; ++s) rows[s] = (int)(((uint64_t)(s + 1) * 2654435761ull) % h.N);
        rows[0] = 0; rows[1] = h.N - 1;
        int* drows; float* ref; CK(hipMalloc(&drows, S * 4)); CK(hipMalloc(&ref, (size_t)S * h.K * 4));
        CK(hipMemcpy(drows, rows.data(), S * 4, hipMemcpyHostToDevice));
        dim3 g((h.K + 127) / 128, S);
        ref_tn_kernel<<<g, 128, 0, st>>>(A[i], B[i], drows, ref, h.M, h.N, h.K);
        CK(hipStreamSynchronize(st));
        std::vector<float> href((size_t)S * h.K), hg(h.K), hgb(h.N);
        CK(hipMemcpy(href.data(), ref, (size_t)S * h.K * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hgb.data(), gb[i], h.N * 4, hipMemcpyDeviceToHost));
        std::vector<uint16_t> ha((size_t)h.M * h.N); CK(hipMemcpy(ha.data(), A[i], (size_t)h.M * h.N * 2, hipMemcpyDeviceToHost));
        const int brows = (int)prob[16 * i + 6];
        int pbad = 0; double maxerr = 0;
        for (int s = 0; s < S; ++s) {
            CK(hipMemcpy(hg.data(), G[i] + (int64_t)rows[s] * h.K, h.K * 4, hipMemcpyDeviceToHost));
            for (int k = 0; k < h.K; ++k) {
                const double r = href[(size_t)s * h.K + k], err = fabs(r - hg[k]);
                if (err > maxerr) maxerr = err;
                if (!(err <= 0.02 + 2e-3 * fabs(r))) ++pbad;
            }
            double r = 0; for (int m = 0; m < brows; ++m) r += bf2f(ha[(size_t)m * h.N + rows[s]]);
            if (!(fabs(r - hgb[rows[s]]) <= 0.05 + 2e-3 * fabs(r))) { if (pbad < 3) printf("   bias mismatch problem %d n %d: ref %g got %g\n", i, rows[s], r, hgb[rows[s]]); ++pbad; }
        }
        printf("  problem %2d (%5d x %4d x %4d): maxerr %.4f %s\n", i, h.M, h.N, h.K, maxerr, pbad ? "FAIL" : "ok");
        bad += pbad;
        CK(hipFree(drows)); CK(hipFree(ref));
    }
    float ms;
    for (int i = 0; i < 2; ++i) grouped();
    CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) grouped(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("grouped : %8.1f us per step-equivalent  %7.1f TF/s\n", ms * 1e3, flops / ms / 1e9);
    for (int i = 0; i < 2; ++i) single();
    CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) single(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("per-layer: %8.1f us per step-equivalent  %7.1f TF/s\n", ms * 1e3, flops / ms / 1e9);
    if (getenv("LAB_PROF")) {
        unsigned long long* prof; CK(hipMalloc(&prof, 4096 * 32)); CK(hipMemset(prof, 0, 4096 * 32));
        ase_hip_debug_nt_profile(prof); grouped(); CK(hipStreamSynchronize(st)); ase_hip_debug_nt_profile(nullptr);
        print_profile(prof, 4096);
    }
    printf("%s\n", bad ? "GROUPED FAIL" : "grouped ok");
    return bad ? 4 : 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "tng")) return run_grouped(argc > 2 ? atoi(argv[2]) : 10);
    if (argc < 5) { printf("usage: gemm_lab nt|tn M N K [reps] [aux] [relu]  |  gemm_lab tng [reps]\n"); return 1; }
    const bool tn = !strcmp(argv[1], "tn");
    const int M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]);
    const int reps = argc > 5 ? atoi(argv[5]) : 20;
    const int use_aux = argc > 6 ? atoi(argv[6]) : 0, relu = argc > 7 ? atoi(argv[7]) : 1;
    const int S = 48;
    hipStream_t st; CK(hipStreamCreate(&st));

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<int> rows(S);
    const int rdim = tn ? N : M;
    for (int s = 0; s < S; ++s) rows[s] = (int)(((uint64_t)(s + 1) * 2654435761ull) % rdim);
    rows[0] = 0; rows[1] = rdim - 1; if (rdim > 300) { rows[2] = 255; rows[3] = 256; rows[4] = 127; rows[5] = 128; }
    int* drows; CK(hipMalloc(&drows, S * 4)); CK(hipMemcpy(drows, rows.data(), S * 4, hipMemcpyHostToDevice));
    if (!tn) {
        bf16_t *A, *B, *C, *aux = nullptr; float* bias; float* ref;
        CK(hipMalloc(&A, (int64_t)M * K * 2)); CK(hipMalloc(&B, (int64_t)N * K * 2)); CK(hipMalloc(&C, (int64_t)M * N * 2));
        CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&ref, (int64_t)S * N * 4));
        fill_kernel<<<1024, 256, 0, st>>>(A, (int64_t)M * K, 1, 1.0f);
        fill_kernel<<<1024, 256, 0, st>>>(B, (int64_t)N * K, 2, 0.05f);
        fillf_kernel<<<(N + 255) / 256, 256, 0, st>>>(bias, N, 3, 0.5f);
        uint32_t* bitsbuf = nullptr;     // use_aux: 1 = bf16 mask, 2 = bit mask (consumer), 3 = produce mask_out
        CK(hipMalloc(&bitsbuf, (int64_t)M * ((N + 31) / 32) * 4));
        if (use_aux) { CK(hipMalloc(&aux, (int64_t)M * N * 2)); fill_kernel<<<1024, 256, 0, st>>>(aux, (int64_t)M * N, 4, 1.0f); }
        if (use_aux == 2) pack_bits_kernel<<<4096, 256, 0, st>>>(aux, bitsbuf, M, N);
        CK(hipMemsetAsync(C, 0xff, (int64_t)M * N * 2, st));
        auto run = [&]() {
            int rc = ase_hip_gemm_nt(A, K, B, K, C, N, bias, use_aux == 2 ? (void*)bitsbuf : (void*)aux, use_aux == 2 ? (N + 31) / 32 : N, 0, 0, nullptr, 0,
                                     use_aux == 3 ? bitsbuf : nullptr, (N + 31) / 32, M, N, K, relu ? ASE_ACT_RELU : ASE_ACT_NONE,
                                     use_aux == 2 ? ASE_AUX_RELU_BITS : (use_aux == 1 ? ASE_AUX_RELU_MASK : ASE_AUX_NONE), 0, 1.0f, nullptr, ASE_BF16, st);
            if (rc) { printf("gemm_nt failed: %s\n", ase_hip_last_error()); exit(3); }
        };
        run();
        dim3 g((N + 127) / 128, S);
        ref_nt_kernel<<<g, 128, 0, st>>>(A, B, bias, (use_aux == 1 || use_aux == 2) ? aux : nullptr, drows, ref, N, K, relu);
        CK(hipStreamSynchronize(st));
        std::vector<float> href((size_t)S * N); std::vector<uint16_t> hc(N);
        CK(hipMemcpy(href.data(), ref, (size_t)S * N * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0; int bad = 0;
        for (int s = 0; s < S; ++s) {
            CK(hipMemcpy(hc.data(), C + (int64_t)rows[s] * N, N * 2, hipMemcpyDeviceToHost));
            for (int n = 0; n < N; ++n) {
                const double r = href[(size_t)s * N + n], c = bf2f(hc[n]);
                const double err = fabs(r - c);
                if (err > maxerr) maxerr = err;
                if (fabs(r) > maxref) maxref = fabs(r);
                if (!(err <= 0.02 + 0.01 * fabs(r))) { if (bad < 5) printf("  mismatch row %d col %d: ref %g got %g\n", rows[s], n, r, c); ++bad; }
            }
        }
        if (use_aux == 3) {     // the produced bit mask must equal (stored C > 0) on the sampled rows
            std::vector<uint32_t> hb((N + 31) / 32);
            for (int s = 0; s < S; ++s) {
                CK(hipMemcpy(hc.data(), C + (int64_t)rows[s] * N, N * 2, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hb.data(), bitsbuf + (int64_t)rows[s] * ((N + 31) / 32), hb.size() * 4, hipMemcpyDeviceToHost));
                for (int n = 0; n < N; ++n)
                    if (((hb[n >> 5] >> (n & 31)) & 1u) != (bf2f(hc[n]) > 0.f ? 1u : 0u)) { if (bad < 5) printf("  mask bit mismatch row %d col %d\n", rows[s], n); ++bad; }
            }
        }
        for (int i = 0; i < 3; ++i) run();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        if (getenv("LAB_PROF")) {
            const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
            unsigned long long* prof; CK(hipMalloc(&prof, tiles * 32)); CK(hipMemset(prof, 0, tiles * 32));
            ase_hip_debug_nt_profile(prof);
            run(); CK(hipStreamSynchronize(st));
            ase_hip_debug_nt_profile(nullptr);
            print_profile(prof, tiles);
        }
        printf("NT %6d x %5d x %5d aux=%d: %8.1f us %8.1f TF/s   maxerr %.4f (max|ref| %.2f) bad %d %s\n", M, N, K, use_aux, ms * 1e3,
               2.0 * M * N * K / ms / 1e9, maxerr, maxref, bad, bad ? "FAIL" : "ok");
    } else {
        bf16_t *A, *B; float *G, *ref, *gb;
        CK(hipMalloc(&A, (int64_t)M * N * 2)); CK(hipMalloc(&B, (int64_t)M * K * 2)); CK(hipMalloc(&G, (int64_t)N * K * 4));
        CK(hipMalloc(&gb, N * 4)); CK(hipMemsetAsync(gb, 0, N * 4, st));
        CK(hipMalloc(&ref, (int64_t)S * K * 4));
        fill_kernel<<<1024, 256, 0, st>>>(A, (int64_t)M * N, 1, 1.0f);
        fill_kernel<<<1024, 256, 0, st>>>(B, (int64_t)M * K, 2, 0.05f);
        CK(hipMemsetAsync(G, 0, (int64_t)N * K * 4, st));
        auto run = [&]() {
            int rc = ase_hip_gemm_tn(A, N, B, K, G, use_aux ? gb : nullptr, 0, M, N, K, N, K, K, K, 1.0f, nullptr, ASE_BF16, st);
            if (rc) { printf("gemm_tn failed: %s\n", ase_hip_last_error()); exit(3); }
        };
        run();
        dim3 g((K + 127) / 128, S);
        ref_tn_kernel<<<g, 128, 0, st>>>(A, B, drows, ref, M, N, K);
        CK(hipStreamSynchronize(st));
        std::vector<float> href((size_t)S * K), hg(K);
        CK(hipMemcpy(href.data(), ref, (size_t)S * K * 4, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0; int bad = 0;
        for (int s = 0; s < S; ++s) {
            CK(hipMemcpy(hg.data(), G + (int64_t)rows[s] * K, K * 4, hipMemcpyDeviceToHost));
            for (int k = 0; k < K; ++k) {
                const double r = href[(size_t)s * K + k], c = hg[k], err = fabs(r - c);
                if (err > maxerr) maxerr = err;
                if (fabs(r) > maxref) maxref = fabs(r);
                if (!(err <= 0.02 + 2e-3 * fabs(r))) { if (bad < 5) printf("  mismatch n %d k %d: ref %g got %g\n", rows[s], k, r, c); ++bad; }
            }
        }
        if (use_aux) {      // bias gradient = column sums of A, checked on the host for the sampled columns
            std::vector<float> hgb(N); CK(hipMemcpy(hgb.data(), gb, N * 4, hipMemcpyDeviceToHost));
            std::vector<uint16_t> ha((size_t)M * N); CK(hipMemcpy(ha.data(), A, (size_t)M * N * 2, hipMemcpyDeviceToHost));
            for (int s = 0; s < S; ++s) {
                double r = 0; for (int m = 0; m < M; ++m) r += bf2f(ha[(size_t)m * N + rows[s]]);
                if (!(fabs(r - hgb[rows[s]]) <= 0.05 + 2e-3 * fabs(r))) { if (bad < 5) printf("  bias mismatch n %d: ref %g got %g\n", rows[s], r, hgb[rows[s]]); ++bad; }
            }
        }
        for (int i = 0; i < 3; ++i) run();
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) run();
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        if (getenv("LAB_PROF")) {
            const int tiles = 4096;
            unsigned long long* prof; CK(hipMalloc(&prof, tiles * 32)); CK(hipMemset(prof, 0, tiles * 32));
            ase_hip_debug_nt_profile(prof);
            run(); CK(hipStreamSynchronize(st));
            ase_hip_debug_nt_profile(nullptr);
            print_profile(prof, tiles);
        }
        printf("TN %6d x %5d x %5d: %8.1f us %8.1f TF/s   maxerr %.4f (max|ref| %.2f) bad %d %s\n", M, N, K, ms * 1e3,
               2.0 * M * N * K / ms / 1e9, maxerr, maxref, bad, bad ? "FAIL" : "ok");
    }
    return 0;
}
