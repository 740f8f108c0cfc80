// Micro-benchmark (round 6): what does one LDS-DMA piece (64 lanes x 16 B = 1 KiB, HBM/L2 -> LDS) cost the wave that issues it, and how
// many can a CU retire per cycle - by instruction form (buffer_load ... lds through an SGPR descriptor vs global_load_lds with 64-bit
// per-lane addresses), by the number of waves issuing at once, and with 0 / 2 / 4 MFMAs (32 x 32 x 16 bf16, 32 cycles each) between two
// pieces.  One 256-thread workgroup per CU (128 KiB of LDS), the access pattern of the 256 x 256 x 64 NT kernels: 8 rows x 128 B per
// piece out of a row-major [rows][K] bf16 matrix with a 2048-byte pitch, 16 pieces per wave and K-tile, one K-tile kept in flight.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dma_bench dma_bench.hip && ./dma_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int IMM> __device__ __forceinline__ void dma_buf(uint32_t lds_s, uint32_t voff, i32x4 rs, uint32_t soff) {
    asm volatile("s_add_u32 m0, %0, %1\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                 :
                 : "s"(lds_s), "n"(IMM), "v"(voff), "s"(rs), "s"(soff)
                 : "memory", "scc");
}

// MODE 0: buffer_load lds (asm), 1: global_load_lds builtin.  NMF: MFMAs between two pieces.
template <int MODE, int NMF>
__global__ __launch_bounds__(256) void dma_kernel(const char* A, int64_t lda, int nk, int reps, int nwaves, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (wid >= nwaves) return;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const uint64_t a = (uint64_t)(uintptr_t)A;
    const i32x4 rs = i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)0x7FFFFFFF, 0x00020000};
    uint32_t voff[16];
    const char* gp[16];
    const int lr = lane >> 3, slot = lane & 7;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int r = (q * 4 + wid) * 8 + lr;                 // 512 rows per workgroup (A and B halves of a K-tile image)
        const int64_t grow = (int64_t)blockIdx.x * 512 + r;
        const uint32_t ch = (uint32_t)((slot ^ ((r >> 1) & 7)) << 4);
        voff[q] = (uint32_t)(grow * lda) + ch;
        gp[q] = A + grow * lda + ch;
    }
    uint32_t dst = __builtin_amdgcn_readfirstlane(lds0 + wid * 1024);
    f32x16 c0, c1;
#pragma unroll
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    bf16x8 x, y;
#pragma unroll
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(float)(lane + e); y[e] = (__bf16)0.5f; }
    asm volatile("s_nop 4" ::: "memory");
    __syncthreads();
    const long long t0 = clock64();
    uint32_t soff = 0;
    int kt = 0;
    for (int it = 0; it < reps; ++it) {
#define PIECE(q)                                                                                                          \
        do {                                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
            if constexpr (MODE == 0) dma_buf<(q) * 4096>(dst, voff[q], rs, soff);                                             \
            else __builtin_amdgcn_global_load_lds((gptr_t*)(gp[q] + soff), (lptr_t*)(uintptr_t)(dst + (q) * 4096), 16, 0, 0);  \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
            if constexpr (NMF >= 1) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0);                          \
            if constexpr (NMF >= 2) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0);                          \
            if constexpr (NMF >= 3) c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c0, 0, 0, 0);                          \
            if constexpr (NMF >= 4) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c1, 0, 0, 0);                          \
            __builtin_amdgcn_sched_barrier(0);                                                                                \
        } while (0)
        PIECE(0); PIECE(1); PIECE(2); PIECE(3); PIECE(4); PIECE(5); PIECE(6); PIECE(7);
        PIECE(8); PIECE(9); PIECE(10); PIECE(11); PIECE(12); PIECE(13); PIECE(14); PIECE(15);
#undef PIECE
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        dst ^= 65536u;
        soff += 128;
        if (++kt == nk) { kt = 0; soff = 0; }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    if (lane == 0) {
        out[(blockIdx.x * 4 + wid) * 2] = (unsigned long long)(t1 - t0);
        out[(blockIdx.x * 4 + wid) * 2 + 1] = (unsigned long long)(s != 12345.f);
    }
}

template <int MODE, int NMF> static void run(const char* A, int64_t lda, int nk, int grid, int nwaves, unsigned long long* dout) {
    const int reps = 64;
    auto kern = dma_kernel<MODE, NMF>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    kern<<<grid, 256, 131072>>>(A, lda, nk, reps, nwaves, dout);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    kern<<<grid, 256, 131072>>>(A, lda, nk, reps, nwaves, dout);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(grid * 8);
    CK(hipMemcpy(h.data(), dout, grid * 64, hipMemcpyDeviceToHost));
    double cyc = 0; int n = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < nwaves; ++w) { cyc += (double)h[(b * 4 + w) * 2]; ++n; }
    cyc /= n;
    const double pieces = 16.0 * reps;
    printf("%-10s MFMAs/piece %d  grid %3d  waves %d : %7.1f cycles per piece per wave, %6.1f B/clk/CU, kernel %.1f us, %.2f TB/s chip, clock %.2f GHz\n",
           MODE == 0 ? "buffer_lds" : "global_lds", NMF, grid, nwaves, cyc / pieces, 1024.0 * nwaves * pieces / cyc, ms * 1e3,
           1024.0 * nwaves * pieces * grid / (ms * 1e-3) / 1e12, cyc / (ms * 1e-3) / 1e9);
}

int main() {
    const int64_t K = 1024, lda = K * 2;
    const int64_t rows = 256 * 512;
    char* A; CK(hipMalloc(&A, rows * lda)); CK(hipMemset(A, 1, rows * lda));
    unsigned long long* dout; CK(hipMalloc(&dout, 256 * 64));
    const int nk = 16;
    for (int grid : {1, 256}) {
        for (int nw : {1, 2, 4}) {
            run<0, 0>(A, lda, nk, grid, nw, dout);
            run<1, 0>(A, lda, nk, grid, nw, dout);
        }
        run<0, 2>(A, lda, nk, grid, 4, dout);
        run<1, 2>(A, lda, nk, grid, 4, dout);
        run<0, 4>(A, lda, nk, grid, 4, dout);
        run<1, 4>(A, lda, nk, grid, 4, dout);
    }
    return 0;
}
