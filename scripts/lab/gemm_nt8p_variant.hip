// LAB VARIANT, NOT BUILT (round 3): the phased NT kernel with the WEIGHTS read from a packed MFMA-fragment copy straight into
// registers (no LDS for B, four-tile A ring, one barrier per K-tile).  Measured (gpurun r3e, scripts/lab/r3e.sh):
// 16384 x 1024 x 1024 41.4 us against 38.4 us of gemm_nt8_kernel (main loop 29.3 vs 25.7 us), 8192^3 942 vs 1135 TF/s - the
// 8 direct fragment loads per wave and K-tile cost more than the LDS round trip they replace - and its hand-counted vmcnt
// scheme still had a race (intermittent wrong tiles).  The bound it chased: with B costing NOTHING (ASE_NT8_V=320 ablation)
// the kernel runs 33.5 us, i.e. at most 13 % at K = 1024.  Dropped; kept for the record (it needs csrc/gemm_nt.h).
// The phased NT kernel with the weights read from their packed copy (see the comment block below) + the pack kernels.
#include "gemm_nt.h"

using namespace ase_nt;

namespace {

// ------------------------------------------------------------------------------------------------
// NT, phased 256 x 256 tile with the B operand (the WEIGHTS) taken from a packed copy straight into registers ("nt8p").
// Why: in the kernel above the data path, not the matrix pipe, sets the pace - per K-tile and CU the LDS moves 64 KB of
// DMA writes + 192 KB of fragment reads (ablations: the loop without MFMAs still takes 81 % of its time), and only
// 48-64 KB of loads are in flight.  Weights are ours to lay out: ase_hip_pack_b stores them as 1-KiB chunks
// [n / 32][k / 16][lane][8 k-values] = exactly what one lane of a 32 x 32 x 16 MFMA B fragment holds, so ONE fully
// coalesced global_load_dwordx4 per (32 output columns, 16 k-values) brings a fragment from L2 into registers: no LDS
// write, no LDS read, no bank pattern for B.  LDS then carries A only (32 KB per K-tile): a FOUR-tile ring, A prefetched
// three K-tiles ahead and B one K-tile ahead in a second register set - ~100-190 KB in flight per CU, LDS traffic per K-tile
// 160 KB instead of 256 KB - and ONE barrier per K-tile instead of eight (the waves of a workgroup drift apart by
// themselves between barriers: while one wave of a SIMD waits for its fragments its partner multiplies).
//   per K-tile t and wave:  s_waitcnt vmcnt(4) [B(t) and A(t) have landed; A(t + 2) may still fly] | s_barrier |
//       issue B(t + 1) (8 loads) and A(t + 3) (4 DMA pieces into the slot A(t - 1) left) |
//       read A row blocks 0, 1 | 16 MFMAs | read A row blocks 2, 3 | 16 MFMAs
// Loads and LDS reads are inline asm with hand-counted waits (behind the builtins hipcc drains vmcnt in front of every LDS
// read that follows an LDS-DMA).  Same wave tiling, accumulator layout and row-per-lane epilogue as gemm_nt8_kernel (SW).
// ------------------------------------------------------------------------------------------------
template <int OFF> __device__ __forceinline__ void lds_read128(i32x4& f, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF) : "memory");
}
// 16 bytes per lane from (wave-uniform base in SGPRs) + (32-bit per-lane offset) + immediate
template <int OFF> __device__ __forceinline__ void gload128(i32x4& f, uint32_t voff, const char* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(f) : "v"(voff), "s"(sbase), "n"(OFF) : "memory");
}
__device__ __forceinline__ void retire_lgkm(i32x4 (&a)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : : "memory");
}
template <int N> __device__ __forceinline__ void retire_vm(i32x4 (&b)[2][4]) {
    asm volatile("s_waitcnt vmcnt(%8)"
                 : "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[0][3]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2]), "+v"(b[1][3])
                 : "n"(N) : "memory");
}
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const char*)(((uint64_t)hi << 32) | lo);
}

// per-wave state of the loop: everything wave-uniform lives in SGPRs, per lane only a handful of 32-bit offsets
struct NT8P {
    const char* a_base;        // uniform: A + (bm0 + wid * 32) * lda
    const char* b_base;        // uniform: packed chunks of the wave's first fragment
    int64_t lda8;              // uniform: 8 rows of A in bytes
    int64_t b_frag;            // uniform: bytes between the wave's two fragments (one n-tile of chunks)
    uint32_t a_voff[2];        // per lane: row (lane >> 3) * lda + swizzled 16-byte chunk, for even / odd 8-row pieces
    uint32_t b_voff;           // per lane: lane * 16
    uint32_t roff[4];          // per lane: LDS byte address of the fragment reads of row block 0 (ring slot 0), per k-step
    char* a_dst;               // uniform: LDS address of the wave's 32 rows in ring slot 0
};

template <typename T>
__device__ __forceinline__ void nt8p_issue_a(const NT8P& w, int t) {
    constexpr int RB = 128, kSlot = 256 * RB;
    char* d = w.a_dst + (t & 3) * kSlot;
    const char* s = w.a_base + (int64_t)t * 128;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        // the address is (wave-uniform 64-bit base in SGPRs) + (32-bit per-lane offset): the empty asm keeps the compiler
        // from re-associating it into four loop-invariant 64-bit per-lane pointers (8 VGPRs it then has to spill)
        const char* sg = s + g * w.lda8;
        asm volatile("" : "+s"(sg));
        __builtin_amdgcn_global_load_lds((gptr_t*)(sg + w.a_voff[g & 1]), (lptr_t*)(d + g * 8 * RB), 16, 0, 0);
    }
}
__device__ __forceinline__ void nt8p_issue_b(const NT8P& w, int t, i32x4 (&b)[2][4]) {
    const char* q0 = w.b_base + (int64_t)t * 4096;
    const char* q1 = q0 + w.b_frag;
    gload128<0>(b[0][0], w.b_voff, q0); gload128<1024>(b[0][1], w.b_voff, q0);
    gload128<2048>(b[0][2], w.b_voff, q0); gload128<3072>(b[0][3], w.b_voff, q0);
    gload128<0>(b[1][0], w.b_voff, q1); gload128<1024>(b[1][1], w.b_voff, q1);
    gload128<2048>(b[1][2], w.b_voff, q1); gload128<3072>(b[1][3], w.b_voff, q1);
}
template <int BLK> __device__ __forceinline__ void nt8p_read(i32x4 (&a)[4], const NT8P& w, uint32_t so) {
    lds_read128<BLK * 32 * 128>(a[0], w.roff[0] + so);
    lds_read128<BLK * 32 * 128>(a[1], w.roff[1] + so);
    lds_read128<BLK * 32 * 128>(a[2], w.roff[2] + so);
    lds_read128<BLK * 32 * 128>(a[3], w.roff[3] + so);
}
template <typename T> __device__ __forceinline__ void nt8p_mma(f32x16& c0, f32x16& c1, const i32x4 (&a)[4], const i32x4 (&b)[2][4]) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        c0 = nt8_mfma<T, true>(a[ks], b[0][ks], c0);
        c1 = nt8_mfma<T, true>(a[ks], b[1][ks], c1);
    }
}

// one K-tile: B(t) = bc (landed), prefetch B(t + 1) into bn.  One A row block (4 fragment registers sets) at a time: the
// register budget (128 accumulators + 2 x 32 for B) leaves room for 16 - the LDS round trip of a block is covered by the
// SIMD's other wave, which is never in step with this one between barriers.
template <typename T>
__device__ __forceinline__ void nt8p_tile(const NT8P& w, int t, int nk, f32x16 (&acc)[4][2], i32x4 (&bc)[2][4], i32x4 (&bn)[2][4],
                                          i32x4 (&x)[4]) {
    // B(t) and A(t) have landed for this wave; younger in its queue: A(t + 2) (t = 0: A(1), A(2))
    if (t == 0) {
        if (nk > 2) retire_vm<8>(bc); else if (nk > 1) retire_vm<4>(bc); else retire_vm<0>(bc);
    } else if (t + 2 < nk) retire_vm<4>(bc);
    else retire_vm<0>(bc);
    NT8_BARRIER();                                   // ... for every wave; and every wave is done reading A(t - 1)
    const uint32_t so = (uint32_t)(t & 3) * (256 * 128);
    nt8p_read<0>(x, w, so);
    if (t + 1 < nk) nt8p_issue_b(w, t + 1, bn);
    if (t + 3 < nk) nt8p_issue_a<T>(w, t + 3);
    retire_lgkm(x);
    __builtin_amdgcn_s_setprio(1);
    nt8p_mma<T>(acc[0][0], acc[0][1], x, bc);
    __builtin_amdgcn_s_setprio(0);
    nt8p_read<1>(x, w, so);
    retire_lgkm(x);
    __builtin_amdgcn_s_setprio(1);
    nt8p_mma<T>(acc[1][0], acc[1][1], x, bc);
    __builtin_amdgcn_s_setprio(0);
    nt8p_read<2>(x, w, so);
    retire_lgkm(x);
    __builtin_amdgcn_s_setprio(1);
    nt8p_mma<T>(acc[2][0], acc[2][1], x, bc);
    __builtin_amdgcn_s_setprio(0);
    nt8p_read<3>(x, w, so);
    retire_lgkm(x);
    __builtin_amdgcn_s_setprio(1);
    nt8p_mma<T>(acc[3][0], acc[3][1], x, bc);
    __builtin_amdgcn_s_setprio(0);
}

template <typename T, int V>
__global__ __launch_bounds__(512) void gemm_nt8p_kernel(NTParams p) {
    static_assert(sizeof(T) == 2, "16-bit storage types");
    constexpr int RB = 128, BM = 256, BN = 256, kSlot = BM * RB;         // 32 KiB = one K-tile of A
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int bm0 = (tile / p.tiles_n) * BM, bn0 = (tile % p.tiles_n) * BN;
    const int nk = p.K / 64;
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 0] = wall_clock64();

    NT8P w;
    {
        // A: per K-tile every wave moves 4 pieces of 8 rows (1 KiB each): rows wid * 32 + 8 g + (lane >> 3); M % 256 == 0.
        // LDS slot of chunk c of row r: c ^ ((r >> 1) & 7), and ((8 g + lr) >> 1) & 7 = (lr >> 1) ^ 4 (g & 1)
        const int lr = lane >> 3, slot = lane & 7;
        w.a_base = uniform_ptr(p.A + (int64_t)(bm0 + wid * 32) * p.lda);
        w.lda8 = 8 * p.lda;
        w.a_voff[0] = (uint32_t)(lr * p.lda) + ((slot ^ (lr >> 1)) << 4);
        w.a_voff[1] = (uint32_t)(lr * p.lda) + ((slot ^ (lr >> 1) ^ 4) << 4);
        w.a_dst = smem + (wid * 32) * RB;
        // B: packed chunks [n-tile][K / 16][64 lanes][16 B]; the wave's fragments are n-tiles (bn0 + 64 wc) / 32 and the next
        // (the packed buffer is padded to whole 256-row tiles: no clamp)
        const int64_t kc = p.K / 16;
        w.b_frag = kc * 1024;
        w.b_base = uniform_ptr(p.Bp + (int64_t)((bn0 + wc * 64) >> 5) * w.b_frag);
        w.b_voff = lane * 16;
        const int r = lane & 31, h = lane >> 5, sw = lds_swz<RB>(r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w.roff[ks] = (uint32_t)(uintptr_t)smem + wr * 128 * RB + r * RB + (((ks * 2 + h) ^ sw) << 4);
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // prologue: the wave tile's mask words (oldest in the queue), B(0), A(0), A(1), A(2)
    const bool mask_dma = p.aux_mode == ASE_AUX_RELU_BITS && bn0 + wc * 64 < p.N;
    if (mask_dma) {
        char* mlds = smem + 4 * kSlot + wid * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = bm0 + wr * 128 + i * 32 + (lane & 31);
            const int ma = (m >= p.aux_split) ? m - p.aux_delta : m;
            const uint32_t* mw = reinterpret_cast<const uint32_t*>(p.aux + (int64_t)min(ma, p.M - 1) * p.ldaux) +
                                 ((bn0 + wc * 64) >> 5) + (lane >> 5);
            __builtin_amdgcn_global_load_lds((gptr_t*)mw, (lptr_t*)(mlds + i * 256), 4, 0, 0);
        }
    }
    i32x4 bA[2][4], bB[2][4], x[4];
    nt8p_issue_b(w, 0, bA);
    nt8p_issue_a<T>(w, 0);
    if (nk > 1) nt8p_issue_a<T>(w, 1);
    if (nk > 2) nt8p_issue_a<T>(w, 2);
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 1] = wall_clock64();
    int t = 0;
    for (; t + 1 < nk; t += 2) {                   // the two register sets of B swap roles every K-tile
        nt8p_tile<T>(w, t, nk, acc, bA, bB, x);
        nt8p_tile<T>(w, t + 1, nk, acc, bB, bA, x);
    }
    if (t < nk) nt8p_tile<T>(w, t, nk, acc, bA, bB, x);
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 2] = wall_clock64();

    uint32_t row_bits[4][2];
    if (mask_dma) {
        const uint32_t* mw = reinterpret_cast<const uint32_t*>(smem + 4 * kSlot + wid * 1024);      // (wave-private, retired long ago)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            row_bits[i][0] = mw[i * 64 + (lane & 31)];
            row_bits[i][1] = mw[i * 64 + 32 + (lane & 31)];
        }
        nt8_epilogue_rows<T, 2>(p, acc, lane, bm0 + wr * 128, bn0 + wc * 64, row_bits);
    } else nt8_epilogue_rows<T, 0>(p, acc, lane, bm0 + wr * 128, bn0 + wc * 64, row_bits);
    if (p.prof) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) p.prof[blockIdx.x * 4 + 3] = wall_clock64();
    }
}

template <typename T, int V> int launch_nt8p(const NTParams& p0, unsigned long long* prof, hipStream_t stream) {
    constexpr int lds = 4 * 256 * 128 + 8 * 1024;       // four K-tiles of A + 1 KiB of mask words per wave
    static bool attr_done = false;
    auto kern = gemm_nt8p_kernel<T, V>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            ase_set_error("gemm_nt8p: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return ASE_ELAUNCH;
        }
        attr_done = true;
    }
    NTParams p = p0;
    p.prof = prof;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    ASE_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), lds, stream, p);
    ASE_CHECK_LAUNCH("gemm_nt8p");
    return ASE_OK;
}

// B[N, K] (row-major, 16-bit) -> packed fragment chunks (see gemm_nt8p_kernel): chunk (nt, kc) = 64 lanes x 16 bytes, lane
// (r = l & 31, h = l >> 5) holds B[32 nt + r][16 kc + 8 h .. + 7]; n-tiles up to a whole 256-row tile, rows >= N are zero.
// One thread per 16-byte piece.
__global__ __launch_bounds__(256) void pack_b_kernel(const char* __restrict__ B, int64_t ldb, int N, int K, char* __restrict__ Bp) {
    const int64_t kc = K / 16, pieces = (int64_t)((N + 255) / 256 * 8) * kc * 64;      // whole 256-row tiles (zero rows past N)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int64_t c = i >> 6, nt = c / kc, k16 = c - nt * kc;
        const int n = (int)nt * 32 + (lane & 31);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n < N) v = *reinterpret_cast<const uint4*>(B + (int64_t)n * ldb + (k16 * 16 + (lane >> 5) * 8) * 2);
        *reinterpret_cast<uint4*>(Bp + i * 16) = v;
    }
}

// several matrices by ONE launch: desc[i] = {B, ldb (elements), N, K, Bp, 0} (int64 each); blockIdx.y = matrix
__global__ __launch_bounds__(256) void pack_b_multi_kernel(const int64_t* __restrict__ desc) {
    const int64_t* d = desc + 6 * blockIdx.y;
    const char* B = reinterpret_cast<const char*>(d[0]);
    const int64_t ldb = d[1] * 2;
    const int N = (int)d[2], K = (int)d[3];
    char* Bp = reinterpret_cast<char*>(d[4]);
    const int64_t kc = K / 16, pieces = (int64_t)((N + 255) / 256 * 8) * kc * 64;      // whole 256-row tiles (zero rows past N)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < pieces; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const int64_t c = i >> 6, nt = c / kc, k16 = c - nt * kc;
        const int n = (int)nt * 32 + (lane & 31);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (n < N) v = *reinterpret_cast<const uint4*>(B + (int64_t)n * ldb + (k16 * 16 + (lane >> 5) * 8) * 2);
        *reinterpret_cast<uint4*>(Bp + i * 16) = v;
    }
}

// B pointer -> packed copy (registered by the owner of the weights; looked up by ase_hip_gemm_nt on the host)
struct PackedMap {
    static constexpr int kMax = 256;
    const void* key[kMax]; const void* val[kMax]; int64_t ld[kMax]; int n = 0;
};
PackedMap g_packed;
}  // namespace

int ase_nt8p_launch(const NTParams& p, int dtype, unsigned long long* prof, hipStream_t stream) {
    if (dtype == ASE_F16) return launch_nt8p<f16_t, 0>(p, prof, stream);
    return launch_nt8p<bf16_t, 0>(p, prof, stream);
}

const char* ase_packed_lookup(const void* B, int64_t ldb) {
    for (int i = 0; i < g_packed.n; ++i)
        if (g_packed.key[i] == B && g_packed.ld[i] == ldb) return (const char*)g_packed.val[i];
    return nullptr;
}

extern "C" int ase_hip_pack_b(const void* B, int64_t ldb, int N, int K, void* Bp, int dtype, void* stream) {
    ASE_CHECK_ARG(B && Bp && N > 0 && K > 0 && K % 64 == 0 && ldb >= K && ldb % 8 == 0 && ((uintptr_t)B % 16) == 0 &&
                      ((uintptr_t)Bp % 16) == 0, "pack_b: null / misaligned operand or K not a multiple of 64");
    ASE_CHECK_ARG(dtype == ASE_BF16 || dtype == ASE_F16, "pack_b: 16-bit storage types only (dtype %d)", dtype);
    const int64_t pieces = (int64_t)((N + 255) / 256 * 8) * (K / 16) * 64;
    ASE_LAUNCH(pack_b_kernel, dim3((unsigned)((pieces + 255) / 256 > 4096 ? 4096 : (pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
               (const char*)B, ldb * 2, N, K, (char*)Bp);
    ASE_CHECK_LAUNCH("pack_b");
    return ASE_OK;
}

extern "C" int ase_hip_pack_b_multi(const int64_t* desc, int n, int dtype, void* stream) {
    ASE_CHECK_ARG(desc && n > 0, "pack_b_multi: null/empty operand");
    ASE_CHECK_ARG(dtype == ASE_BF16 || dtype == ASE_F16, "pack_b_multi: 16-bit storage types only (dtype %d)", dtype);
    ASE_LAUNCH(pack_b_multi_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, desc);
    ASE_CHECK_LAUNCH("pack_b_multi");
    return ASE_OK;
}

extern "C" int ase_hip_pack_register(const void* B, int64_t ldb, const void* Bp) {
    ASE_CHECK_ARG(B != nullptr, "pack_register: null matrix");
    for (int i = 0; i < g_packed.n; ++i) {
        if (g_packed.key[i] == B) {
            if (Bp) { g_packed.val[i] = Bp; g_packed.ld[i] = ldb; }
            else { g_packed.key[i] = g_packed.key[g_packed.n - 1]; g_packed.val[i] = g_packed.val[g_packed.n - 1];
                   g_packed.ld[i] = g_packed.ld[g_packed.n - 1]; --g_packed.n; }
            return ASE_OK;
        }
    }
    if (!Bp) return ASE_OK;
    ASE_CHECK_ARG(g_packed.n < PackedMap::kMax, "pack_register: table full (%d matrices)", PackedMap::kMax);
    g_packed.key[g_packed.n] = B; g_packed.val[g_packed.n] = Bp; g_packed.ld[g_packed.n] = ldb; ++g_packed.n;
    return ASE_OK;
}

