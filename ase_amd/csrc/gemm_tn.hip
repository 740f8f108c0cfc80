// Weight-gradient ("TN") matrix-core kernels (gfx950 / CDNA4) and the shadow-weight refresh.
//
//   gemm_tn : G += alpha * Aᵀ·B   weight (+ bias) gradient
//   TN kernels: gemm_tn_kernel (128 x 128, register-staged, transposed LDS reads, split over M, f32 atomics) and the phased
//       gemm_tn8 kernels (256 x 256, DMA-staged), single problem or grouped (all weight gradients of a branch in one grid,
//       partial tiles into workspace slabs folded by tn_reduce_kernel).
#include "gemm_nt.h"
#include <stdlib.h>
#include <math.h>
#include <algorithm>
#include <vector>

using namespace ase_nt;

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ void split_bf16(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hi[q] = (bf16_t)x0[q];
        hi[q + 4] = (bf16_t)x1[q];
        lo[q] = (bf16_t)(x0[q] - (float)hi[q]);
        lo[q + 4] = (bf16_t)(x1[q] - (float)hi[q + 4]);
    }
}

// ------------------------------------------------------------------------------------------------
// TN: G[n, k] += alpha * sum_m A[m, n] * B[m, k].  The contraction runs over ROWS of both operands,
// so fragments need the transpose of what a row-major tile holds:
//   bf16: ds_read_b64_tr_b16 (gfx950 LDS transpose read) delivers 4 consecutive m for one column;
//   f32 : the 32x32x2 MFMA takes one scalar per lane, so a plain ds_read_b32 walks a tile row.
// ------------------------------------------------------------------------------------------------
struct TNParams {
    const char* A; int64_t lda;   // bytes
    const char* B; int64_t ldb;   // bytes
    float* G;
    float* gbias;                 // nullable: += column sums of A rows < bias_rows (bias gradient), n < n_real
    int bias_rows;
    int M, N, K;                  // padded widths N (of A), K (of B), in elements
    int n_real, k_real, split_src, split_dst;
    float alpha;
    const float* alpha_dev;       // nullable: device factor multiplied into alpha when the launch runs (1 / the dynamic loss scale)
    int tiles_n, tiles_k, m_chunk;
    unsigned long long* prof;     // debug stamps (ase_hip_debug_nt_profile), else null
};

template <typename T> struct TNGeom;
// bf16 row pitch 256 + 64 B: the 8 (row, 16-column-group) blocks that the 32 lanes of one ds_read_b64_tr_b16 group
// touch land on 8 distinct 32-byte bank slots (pitch = 16 dwords mod 64)
template <> struct TNGeom<bf16_t> { static constexpr int BKM = 64, STRIDE = 256 + 64, CPR = 16; };
template <> struct TNGeom<f16_t> { static constexpr int BKM = 64, STRIDE = 256 + 64, CPR = 16; };
template <> struct TNGeom<float>  { static constexpr int BKM = 16, STRIDE = 512 + 16, CPR = 32; };
template <> struct TNGeom<f32s_t> { static constexpr int BKM = 16, STRIDE = 512 + 16, CPR = 32; };

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

__device__ __forceinline__ bf16x4 lds_tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (lds_bf16x4*)(__attribute__((address_space(3))) void*)(p));
}

template <typename T, int LOADS>
__device__ __forceinline__ void tn_gload(uint4 (&ra)[LOADS], uint4 (&rb)[LOADS], const TNParams& p, int tid, int m0,
                                         int m_end, int bn0, int bk0) {
    constexpr int CPR = TNGeom<T>::CPR, EPC = 16 / (int)sizeof(T);
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int c = tid + kThreads * i;
        const int row = c / CPR, ch = c % CPR;
        const int m = m0 + row;
        const int ca = bn0 + ch * EPC, cb = bk0 + ch * EPC;
        ra[i] = make_uint4(0, 0, 0, 0);
        rb[i] = make_uint4(0, 0, 0, 0);
        if (m < m_end && ca < p.N) ra[i] = *reinterpret_cast<const uint4*>(p.A + (int64_t)m * p.lda + (int64_t)ca * sizeof(T));
        if (m < m_end && cb < p.K) rb[i] = *reinterpret_cast<const uint4*>(p.B + (int64_t)m * p.ldb + (int64_t)cb * sizeof(T));
    }
}
template <typename T, int LOADS>
__device__ __forceinline__ void tn_sstore(const uint4 (&ra)[LOADS], const uint4 (&rb)[LOADS], char* sbuf, int tid) {
    constexpr int CPR = TNGeom<T>::CPR, STRIDE = TNGeom<T>::STRIDE, kOp = TNGeom<T>::BKM * TNGeom<T>::STRIDE;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        const int c = tid + kThreads * i;
        const int row = c / CPR, ch = c % CPR;
        *reinterpret_cast<uint4*>(sbuf + row * STRIDE + ch * 16) = ra[i];
        *reinterpret_cast<uint4*>(sbuf + kOp + row * STRIDE + ch * 16) = rb[i];
    }
}

// running column sums of the staged A chunks (each thread always stages the same 16-byte column chunk)
template <typename T, int LOADS>
__device__ __forceinline__ void tn_colsum(const uint4 (&ra)[LOADS], float (&cs)[8], int tid, int m0, int bias_rows) {
    constexpr int CPR = TNGeom<T>::CPR;
#pragma unroll
    for (int i = 0; i < LOADS; ++i) {
        if (m0 + (tid + kThreads * i) / CPR >= bias_rows) continue;
        if constexpr (sizeof(T) == 2) {
            const typename V16<T>::x8 v = *reinterpret_cast<const typename V16<T>::x8*>(&ra[i]);
#pragma unroll
            for (int q = 0; q < 8; ++q) cs[q] += (float)v[q];
        } else {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&ra[i]);
#pragma unroll
            for (int q = 0; q < 4; ++q) cs[q] += v[q];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void gemm_tn_kernel(TNParams p) {
    using Gm = TNGeom<T>;
    constexpr int BKM = Gm::BKM, STRIDE = Gm::STRIDE, CPR = Gm::CPR;
    constexpr int LOADS = BKM * CPR / kThreads;          // 16-B chunks per thread per operand
    constexpr int kOp = BKM * STRIDE;                    // bytes per operand tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.alpha_dev) p.alpha *= *p.alpha_dev;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wi = wid >> 1, wj = wid & 1;               // wave position in the 128x128 output tile
    const int nwg = p.tiles_n * p.tiles_k;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int bn0 = (tile / p.tiles_k) * 128, bk0 = (tile % p.tiles_k) * 128;
    const int m_begin = blockIdx.z * p.m_chunk;
    const int m_end = min(p.M, m_begin + p.m_chunk);
    if (m_begin >= m_end) return;

    uint4 ra[LOADS], rb[LOADS];
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nt = (m_end - m_begin + BKM - 1) / BKM;
    const bool do_bias = p.gbias != nullptr && bk0 == 0;      // one k-tile column of workgroups also reduces A
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    tn_gload<T, LOADS>(ra, rb, p, tid, m_begin, m_end, bn0, bk0);
    if (do_bias) tn_colsum<T, LOADS>(ra, cs, tid, m_begin, p.bias_rows);
    tn_sstore<T, LOADS>(ra, rb, smem, tid);
    __syncthreads();
    for (int mt = 0; mt < nt; ++mt) {
        const int buf = mt & 1;
        if (mt + 1 < nt) tn_gload<T, LOADS>(ra, rb, p, tid, m_begin + (mt + 1) * BKM, m_end, bn0, bk0);
        const char* sA = smem + buf * 2 * kOp;
        const char* sB = sA + kOp;
        if constexpr (sizeof(T) == 2) {
            // lane l: 16-lane group g = l>>4 -> half h = g>>1 (k-group of the MFMA), column group cg = g&1;
            // within the group lane t supplies the address of row (t>>2), columns (t&3)*4..+3 and receives
            // column t of the 4x16 block (4 consecutive m).
            const int t = lane & 15, g = lane >> 4, h = g >> 1, cg = g & 1;
            const int arow = h * 8 + (t >> 2);
            const int acol = cg * 16 + (t & 3) * 4;
#pragma unroll
            for (int ks = 0; ks < BKM / 16; ++ks) {
                typedef typename V16<T>::x8 x8;
                x8 a[2], b[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const char* pa = sA + (ks * 16 + arow) * STRIDE + ((wi * 2 + f) * 32 + acol) * 2;
                    const char* pb = sB + (ks * 16 + arow) * STRIDE + ((wj * 2 + f) * 32 + acol) * 2;
                    const bf16x4 a0 = lds_tr_read(pa), a1 = lds_tr_read(pa + 4 * STRIDE);
                    const bf16x4 b0 = lds_tr_read(pb), b1 = lds_tr_read(pb + 4 * STRIDE);
                    a[f] = __builtin_bit_cast(x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                    b[f] = __builtin_bit_cast(x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = mfma16<T>(a[i], b[j], acc[i][j]);
            }
        } else if constexpr (std::is_same<T, f32s_t>::value) {
            // one 16-deep step per staged tile (BKM = 16): lane (r, h) gathers rows 8 h .. 8 h + 7 of its column
            const int r = lane & 31, h = lane >> 5;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                f32x4 x0, x1, y0, y1;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x0[q] = *reinterpret_cast<const float*>(sA + (h * 8 + q) * STRIDE + ((wi * 2 + f) * 32 + r) * 4);
                    x1[q] = *reinterpret_cast<const float*>(sA + (h * 8 + 4 + q) * STRIDE + ((wi * 2 + f) * 32 + r) * 4);
                    y0[q] = *reinterpret_cast<const float*>(sB + (h * 8 + q) * STRIDE + ((wj * 2 + f) * 32 + r) * 4);
                    y1[q] = *reinterpret_cast<const float*>(sB + (h * 8 + 4 + q) * STRIDE + ((wj * 2 + f) * 32 + r) * 4);
                }
                split_bf16(x0, x1, ah[f], al[f]);
                split_bf16(y0, y1, bh[f], bl[f]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        } else {
            const int r = lane & 31, h = lane >> 5;
#pragma unroll
            for (int ks = 0; ks < BKM / 2; ++ks) {
                float a[2], b[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    a[f] = *reinterpret_cast<const float*>(sA + (ks * 2 + h) * STRIDE + ((wi * 2 + f) * 32 + r) * 4);
                    b[f] = *reinterpret_cast<const float*>(sB + (ks * 2 + h) * STRIDE + ((wj * 2 + f) * 32 + r) * 4);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
        if (mt + 1 < nt) {
            if (do_bias) tn_colsum<T, LOADS>(ra, cs, tid, m_begin + (mt + 1) * BKM, p.bias_rows);
            tn_sstore<T, LOADS>(ra, rb, smem + (buf ^ 1) * 2 * kOp, tid);
        }
        __syncthreads();
    }

    if (do_bias) {   // block-level reduction over the threads that staged the same column chunk, then one atomic per column
        constexpr int EPC = 16 / (int)sizeof(T);
        float* red = reinterpret_cast<float*>(smem);           // [256][EPC]  (the loop ended with a barrier)
#pragma unroll
        for (int q = 0; q < EPC; ++q) red[tid * EPC + q] = cs[q];
        __syncthreads();
        if (tid < 128) {
            const int ch = tid / EPC, q = tid % EPC;
            float t = 0.f;
            for (int j = ch; j < kThreads; j += CPR) t += red[j * EPC + q];
            const int n = bn0 + tid;
            if (n < p.n_real) atomic_add_f32(p.gbias + n, p.alpha * t);
        }
    }
    const int col_in = lane & 31, row_hi = (lane >> 5) * 4;
    const int gap = p.split_dst - p.split_src;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = bk0 + (wj * 2 + j) * 32 + col_in;
        int kk = -1;
        if (k < p.split_src) kk = k;
        else if (k >= p.split_dst && k - gap < p.k_real) kk = k - gap;
        if (kk < 0) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nbase = bn0 + (wi * 2 + i) * 32 + row_hi;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = nbase + (e & 3) + 8 * (e >> 2);
                if (n < p.n_real) atomic_add_f32(p.G + (int64_t)n * p.k_real + kk, p.alpha * acc[i][j][e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TN, phased 256 x 256 kernel (bf16): the weight-gradient twin of gemm_nt8_kernel.  Output tile 256 (n) x 256 (k), 8
// waves as 2 (n) x 4 (k), contraction over 64 rows m per K-tile.  Both staged tiles are row-major [64 m][512 B] images
// written by the DMA (2 rows per 1-KiB piece) with the 16-byte chunks of row m at slot chunk ^ ((m & 3) << 2): the four
// rows that one 32-lane group of ds_read_b64_tr_b16 touches land in the four 64-byte quarters of the bank window.
//   DMA units (16 KiB = 32 rows of one operand), in issue order: B-lo, A-lo, B-hi, A-hi  (lo / hi = rows 0-31 / 32-63)
//   phase 0: read A-lo (n fragments 0, 1) + B-lo -> acc[0..1][*]      phase 1: A-lo (fragments 2, 3) -> acc[2..3][*]
//   phase 2: A-hi (0, 1) + B-hi                                      phase 3: A-hi (2, 3)
//   issue / wait arithmetic exactly as in the NT kernel (unit 4 t + p + 6 in phase (t, p); waits in phases 1 and 3).
// Bias gradient: workgroups of the first k-tile column multiply one A fragment per wave by a constant all-ones B
// fragment (wave column c owns n fragment c: 4 extra MFMAs per K-tile), so the column sums never leave the matrix pipe.
//
// The split-M partial sums meet in f32 atomics, which run memory-side on this chip (~1.4 TB/s measured: 47 us for the
// 16 x 4 MB of a 1024 x 1024 gradient split 16 ways, against 27 us of main loop).  Hence the GROUPED launch: all weight
// gradients of one optimisation step (they only depend on buffers the data-gradient chain has already written) go
// out as ONE grid whose work items {problem, tile, m range} are sized so that ~256 workgroups each run a long
// contraction (100+ K-tiles): the same 256 x 256 KB of partial sums are then paid once per step, not once per layer.
// ------------------------------------------------------------------------------------------------
constexpr int kTnSlab = 65536 + 256;       // floats per work item in the partial-sum workspace: 256 x 256 tile + 256 bias sums

// (DMA pieces as `buffer_load_dwordx4 ... lds`, round 6 - see NT8Lane in gemm_nt_kernels.h: one 32-bit offset register per piece
//  instead of a 64-bit address, the K-tile's byte offset as the instruction's SGPR offset; the registers it frees are what the
//  small kernels of the step's other branches co-reside in.)
struct TN8Lane {
    uint32_t voff[4][2];       // per-lane byte offset of unit kind (B-lo, A-lo, B-hi, A-hi) x piece from its operand's base, at K-tile 0
    uint32_t dst[4][2];        // wave-uniform LDS byte ADDRESS of the piece in K-tile buffer 0 (buffer 1: + 65536)
    int rbase;                 // per-lane tr-read base: row, 16-byte sub-chunk and half of the lane
    int foffA[4], foffB[2];    // swizzled 64-byte fragment-column offsets
    uint32_t kstep[2];         // bytes per K-tile (64 rows) of B / A
    i32x4 rs[2];               // raw buffer descriptors of B / A (SGPRs)
};

template <int KIND>
__device__ __forceinline__ void tn8_piece(const TN8Lane& L, int g, int tile) {
    const uint32_t lds_s = __builtin_amdgcn_readfirstlane(L.dst[KIND][g] + (uint32_t)(tile & 1) * 65536u);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :
                 : "s"(lds_s), "v"(L.voff[KIND][g]), "s"(L.rs[KIND & 1]), "s"((uint32_t)tile * L.kstep[KIND & 1])
                 : "memory");
}

template <int KIND>
__device__ __forceinline__ void tn8_issue(const TN8Lane& L, char* smem, int tile) {
#pragma unroll
    for (int g = 0; g < 2; ++g) tn8_piece<KIND>(L, g, tile);
}

// fragment (32 columns) x k-step (16 rows): two transposed 8-byte reads = the lane's 8 consecutive m of its column.
// Inline asm on purpose: behind the builtin hipcc drains the DMA queue (vmcnt(0)) in front of every transposed read
// that follows a global_load_lds; the asm reads are ordered by the explicit lgkmcnt(0) + sched_barrier of the phase
// (nt8_sync_in), and the two halves are only joined into the MFMA operand after that wait.
struct tr_pair { bf16x4 lo, hi; };
template <int OFF> __device__ __forceinline__ void tn8_read(tr_pair& f, uint32_t addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(addr), "n"(OFF) : "memory");
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(addr), "n"(OFF + 4 * 512) : "memory");
}
template <typename T> __device__ __forceinline__ typename V16<T>::x8 tn8_join(const tr_pair& f) {      // (raw 16-bit lanes)
    return __builtin_bit_cast(typename V16<T>::x8, __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// 8 MFMAs of one phase (+ 2 for the bias gradient when this wave owns one of the two live A fragments: bias_sel 0 / 1
// picks it with VALU selects, bvec is all ones or - outside bias_rows - all zeros)
// KIND >= 0: the two DMA pieces of unit KIND (K-tile `tile`) are issued after the first and the third MFMA pair of the
// block (see nt8_mma_issue: an LDS-DMA costs ~60 issue cycles beside MFMAs, 100-185 in the read half of a phase)
template <typename T, bool BIAS, int KIND>
__device__ __forceinline__ void tn8_mma(f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11, f32x16& bacc,
                                        const tr_pair (&a)[2][2], const tr_pair (&b)[2][2], int bias_sel,
                                        typename V16<T>::x8 bvec, const TN8Lane& L, char* smem, int tile, bool live) {
    typedef typename V16<T>::x8 x8;
    x8 a0[2], a1[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a0[ks] = tn8_join<T>(a[0][ks]);
        a1[ks] = tn8_join<T>(a[1][ks]);
        const x8 b0 = tn8_join<T>(b[0][ks]), b1 = tn8_join<T>(b[1][ks]);
        c00 = mfma16<T>(a0[ks], b0, c00);
        c10 = mfma16<T>(a1[ks], b0, c10);
        if constexpr (KIND >= 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (live) tn8_piece<KIND>(L, ks, tile);
            __builtin_amdgcn_sched_barrier(0);
        }
        c01 = mfma16<T>(a0[ks], b1, c01);
        c11 = mfma16<T>(a1[ks], b1, c11);
    }
    if (BIAS) {
        if (bias_sel >= 0) {                         // wave-uniform
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const i32x4 x0 = __builtin_bit_cast(i32x4, a0[ks]), x1 = __builtin_bit_cast(i32x4, a1[ks]);
                i32x4 xs;
#pragma unroll
                for (int q = 0; q < 4; ++q) xs[q] = bias_sel ? x1[q] : x0[q];
                bacc = mfma16<T>(__builtin_bit_cast(x8, xs), bvec, bacc);
            }
        }
    }
}

// HALF = 0 / 1: phases 0, 1 (rows 0-31 of the K-tile) / phases 2, 3 (rows 32-63)
template <typename T, bool TAIL, bool BIAS, int HALF>
__device__ __forceinline__ void tn8_half(int t, int nk, const TN8Lane& L, char* smem, const uint32_t (&adA)[4],
                                         const uint32_t (&adB)[2], int bias_frag, typename V16<T>::x8 bvec,
                                         f32x16 (&acc)[4][2], f32x16& bacc) {
    constexpr int RO = HALF * 2 * 8192;           // k-steps 2 HALF, 2 HALF + 1
    const int U = 4 * nk;
    const bool live = !TAIL || (HALF == 0 ? t + 1 < nk : t + 2 < nk);
    const int itile = HALF == 0 ? t + 1 : t + 2;  // K-tile whose units this half issues (B / A of its lo or hi rows)
    tr_pair a[2][2], b[2][2];
    // ---- even phase: A fragments 0, 1 + both B fragments of this half; its DMA unit goes out inside the MFMA block
    tn8_read<32768 + RO>(b[0][0], adB[0]);
    tn8_read<32768 + RO + 8192>(b[0][1], adB[0]);
    tn8_read<32768 + RO>(b[1][0], adB[1]);
    tn8_read<32768 + RO + 8192>(b[1][1], adB[1]);
    tn8_read<RO>(a[0][0], adA[0]);
    tn8_read<RO + 8192>(a[0][1], adA[0]);
    tn8_read<RO>(a[1][0], adA[1]);
    tn8_read<RO + 8192>(a[1][1], adA[1]);
    nt8_sync_in();
    tn8_mma<T, BIAS, HALF == 0 ? 2 : 0>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], bacc, a, b, bias_frag < 2 ? bias_frag : -1,
                                     bvec, L, smem, itile, live);
    nt8_sync_out();
    // ---- odd phase: A fragments 2, 3; the wait retires what the next phase reads (the newest issued unit is the even
    // phase's: three units may stay in flight)
    tn8_read<RO>(a[0][0], adA[2]);
    tn8_read<RO + 8192>(a[0][1], adA[2]);
    tn8_read<RO>(a[1][0], adA[3]);
    tn8_read<RO + 8192>(a[1][1], adA[3]);
    if (!TAIL) wait_dma_units<3>();
    else if (HALF == 0) wait_dma_units_rt(min(U, 4 * t + 7) - (4 * t + 4));
    else if (t + 1 < nk) wait_dma_units_rt(min(U, 4 * t + 9) - (4 * t + 6));
    nt8_sync_in();
    tn8_mma<T, BIAS, HALF == 0 ? 3 : 1>(acc[2][0], acc[2][1], acc[3][0], acc[3][1], bacc, a, b, bias_frag >= 2 ? bias_frag - 2 : -1,
                                     bvec, L, smem, itile, live);
    nt8_sync_out();
}

template <typename T, bool TAIL, bool BIAS>
__device__ __forceinline__ void tn8_ktile(int t, int nk, const TN8Lane& L, char* smem, uint32_t lds0, int bias_frag,
                                          bool bias_on, f32x16 (&acc)[4][2], f32x16& bacc) {
    const uint32_t pa = lds0 + (t & 1) * 65536 + L.rbase;
    uint32_t adA[4], adB[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) adA[i] = pa + L.foffA[i];
#pragma unroll
    for (int j = 0; j < 2; ++j) adB[j] = pa + L.foffB[j];
    typename V16<T>::x8 bvec;
#pragma unroll
    for (int q = 0; q < 8; ++q) bvec[q] = (T)(bias_on ? 1.0f : 0.0f);
    tn8_half<T, TAIL, BIAS, 0>(t, nk, L, smem, adA, adB, bias_frag, bvec, acc, bacc);
    tn8_half<T, TAIL, BIAS, 1>(t, nk, L, smem, adA, adB, bias_frag, bvec, acc, bacc);
}

// one work item: output tile (bn0, bk0) of problem p over the K-tiles [m_begin, m_begin + 64 nk)
// ws / wsb (grouped launch): the work item's slab of the partial-sum workspace - the raw accumulator image (64 K floats,
// chunk ((wave * 8 + i * 2 + j) * 4 + g) x 64 lanes x f32x4: every store instruction writes one contiguous KiB) and 256
// bias partial sums - which tn_reduce_kernel folds into the gradient; null: f32 atomics straight into G.
template <typename T>
__device__ __forceinline__ void tn8_body(const TNParams& p, char* smem, int bn0, int bk0, int m_begin, int nk,
                                         unsigned long long* prof, float* ws = nullptr, float* wsb = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    if (prof && tid == 0) prof[0] = wall_clock64();

    TN8Lane L;
    {
        const int lrow = lane >> 5, slot = lane & 31;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int kind = 0; kind < 4; ++kind) {
                const bool isA = kind & 1;
                const int row0 = (kind >> 1) * 32 + 2 * (g * 8 + wid), row = row0 + lrow;
                int chunk = slot ^ ((row & 3) << 2);
                const int col0 = isA ? bn0 : bk0, width = isA ? p.N : p.K;
                if (col0 + chunk * 8 >= width) chunk = 0;            // columns past the operand: products only reach unstored outputs
                const int64_t ld = isA ? p.lda : p.ldb;
                L.voff[kind][g] = (uint32_t)((int64_t)(m_begin + row) * ld + (int64_t)col0 * 2 + chunk * 16);
                L.dst[kind][g] = (uint32_t)(uintptr_t)smem + (isA ? 0 : 32768) + row0 * 512;
            }
        }
        L.kstep[0] = (uint32_t)(64 * p.ldb);
        L.kstep[1] = (uint32_t)(64 * p.lda);
        const uint64_t pa_ = (uint64_t)(uintptr_t)p.A, pb_ = (uint64_t)(uintptr_t)p.B;
        L.rs[0] = i32x4{(int)(uint32_t)pb_, (int)(uint32_t)(pb_ >> 32), (int)0x7FFFFFFF, 0x00020000};
        L.rs[1] = i32x4{(int)(uint32_t)pa_, (int)(uint32_t)(pa_ >> 32), (int)0x7FFFFFFF, 0x00020000};
        const int t = lane & 15, g4 = lane >> 4, h = g4 >> 1, cg = g4 & 1, s2 = (t >> 2) & 3;
        L.rbase = (h * 8 + (t >> 2)) * 512 + (cg * 2 + ((t & 3) >> 1)) * 16 + (t & 1) * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) L.foffA[i] = ((wr * 4 + i) ^ s2) * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) L.foffB[j] = ((wc * 2 + j) ^ s2) * 64;
    }

    f32x16 acc[4][2], bacc;
#pragma unroll
    for (int e = 0; e < 16; ++e) bacc[e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const bool do_bias = p.gbias != nullptr && bk0 == 0;
    // whole K-tiles below bias_rows contribute to the bias gradient (the host checks bias_rows % 64 == 0)
    const int bias_tiles = do_bias ? max(0, min(nk, (p.bias_rows - m_begin) / 64)) : 0;

    tn8_issue<0>(L, smem, 0);
    tn8_issue<1>(L, smem, 0);
    tn8_issue<2>(L, smem, 0);
    tn8_issue<3>(L, smem, 0);
    if (nk > 1) {
        tn8_issue<0>(L, smem, 1);
        tn8_issue<1>(L, smem, 1);
        wait_dma_units<4>();
    } else {
        wait_dma_units<2>();
    }
    NT8_BARRIER();
    if (prof && tid == 0) prof[1] = wall_clock64();
    if (wr == 1) NT8_BARRIER();

    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;           // LDS byte address of the ring (low half of the flat address)
    int t = 0;
    if (bias_tiles > 0) {
        for (; t + 2 < nk; ++t)
            tn8_ktile<T, false, true>(t, nk, L, smem, lds0, wc, t < bias_tiles, acc, bacc);
        for (; t < nk; ++t)
            tn8_ktile<T, true, true>(t, nk, L, smem, lds0, wc, t < bias_tiles, acc, bacc);
    } else {
        for (; t + 2 < nk; ++t)
            tn8_ktile<T, false, false>(t, nk, L, smem, lds0, wc, false, acc, bacc);
        for (; t < nk; ++t)
            tn8_ktile<T, true, false>(t, nk, L, smem, lds0, wc, false, acc, bacc);
    }
    if (wr == 0) NT8_BARRIER();
    if (prof && tid == 0) prof[2] = wall_clock64();

    const int col_in = lane & 31, row_hi = (lane >> 5) * 4;
    if (ws) {
        f32x4* o = reinterpret_cast<f32x4*>(ws) + lane;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    o[((wid * 8 + i * 2 + j) * 4 + g) * 64] =
                        f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
        if (do_bias && col_in == 0) {              // (zeros when no K-tile of this item lies below bias_rows)
            const int nl = (wr * 4 + wc) * 32 + row_hi;
#pragma unroll
            for (int e = 0; e < 16; ++e) wsb[nl + (e & 3) + 8 * (e >> 2)] = bacc[e];
        }
        if (prof) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) prof[3] = wall_clock64();
        }
        return;
    }
    if (bias_tiles > 0 && col_in == 0) {           // every column of bacc holds the sums: column 0 writes them
        const int nbase = bn0 + (wr * 4 + wc) * 32 + row_hi;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = nbase + (e & 3) + 8 * (e >> 2);
            if (n < p.n_real) atomic_add_f32(p.gbias + n, p.alpha * bacc[e]);
        }
    }
    const int gap = p.split_dst - p.split_src;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = bk0 + (wc * 2 + j) * 32 + col_in;
        int kk = -1;
        if (k < p.split_src) kk = k;
        else if (k >= p.split_dst && k - gap < p.k_real) kk = k - gap;
        if (kk < 0) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int nbase = bn0 + (wr * 4 + i) * 32 + row_hi;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = nbase + (e & 3) + 8 * (e >> 2);
                if (n < p.n_real) atomic_add_f32(p.G + (int64_t)n * p.k_real + kk, p.alpha * acc[i][j][e]);
            }
        }
    }
    if (prof) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) prof[3] = wall_clock64();
    }
}

template <typename T>
__global__ __launch_bounds__(512) void gemm_tn8_kernel(TNParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.alpha_dev) p.alpha *= *p.alpha_dev;
    const int nwg = p.tiles_n * p.tiles_k;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int m_begin = blockIdx.z * p.m_chunk;
    const int m_end = min(p.M, m_begin + p.m_chunk);
    if (m_begin >= m_end) return;
    tn8_body<T>(p, smem, (tile / p.tiles_k) * 256, (tile % p.tiles_k) * 256, m_begin, (m_end - m_begin) / 64,
                p.prof ? p.prof + (blockIdx.z * gridDim.x + blockIdx.x) * 4 : nullptr);
}

// Grouped launch.  problems: device int64[n][16] = {A, lda, B, ldb, G, gbias, bias_rows, M, N, K, n_real, k_real,
// split_src, split_dst, alpha (f32 bits), tiles_k}, leading dimensions in ELEMENTS; work: device int32[n_work][4] =
// {problem, tile, m_begin, nk | slab << 16} (ase_hip_gemm_tn_grouped_plan).
template <typename T>
__global__ __launch_bounds__(512) void gemm_tn8g_kernel(const int64_t* __restrict__ problems,
                                                        const int32_t* __restrict__ work, int n_work,
                                                        unsigned long long* prof, float* __restrict__ ws,
                                                        const float* __restrict__ alpha_dev) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int item = xcd_remap(blockIdx.x, n_work);             // neighbours in the work list share operand panels
    const int32_t* w = work + 4 * item;
    const int pi = __builtin_amdgcn_readfirstlane(w[0]), tile = __builtin_amdgcn_readfirstlane(w[1]);
    const int m_begin = __builtin_amdgcn_readfirstlane(w[2]), w3 = __builtin_amdgcn_readfirstlane(w[3]);
    const int nk = w3 & 0xFFFF, slab = w3 >> 16;               // slab: the item's place in the workspace (reduce table order)
    if (nk == 0) return;                                        // padding of an XCD's position range
    const int64_t* d = problems + 16 * pi;
    TNParams p;
    p.A = reinterpret_cast<const char*>(d[0]); p.lda = d[1] * 2;
    p.B = reinterpret_cast<const char*>(d[2]); p.ldb = d[3] * 2;
    p.G = reinterpret_cast<float*>(d[4]); p.gbias = reinterpret_cast<float*>(d[5]);
    p.bias_rows = (int)d[6]; p.M = (int)d[7]; p.N = (int)d[8]; p.K = (int)d[9];
    p.n_real = (int)d[10]; p.k_real = (int)d[11]; p.split_src = (int)d[12]; p.split_dst = (int)d[13];
    p.alpha = __builtin_bit_cast(float, (int)d[14]) * (alpha_dev ? *alpha_dev : 1.f);
    p.alpha_dev = nullptr;
    p.tiles_k = (int)(d[15] & 0xFFFF);
    tn8_body<T>(p, smem, (tile / p.tiles_k) * 256, (tile % p.tiles_k) * 256, m_begin, nk,
                   prof ? prof + blockIdx.x * 4 : nullptr, ws ? ws + (int64_t)slab * kTnSlab : nullptr,
                   ws ? ws + (int64_t)slab * kTnSlab + 65536 : nullptr);
}

// Second kernel of the grouped launch: G += alpha * (sum of the work items' partial tiles), gbias likewise.  One workgroup
// per (reduce entry, quarter tile); red[r] = {problem, tile, first item, splits}, split s of a tile sits `tiles of the
// problem` items further (ase_hip_gemm_tn_grouped_plan's order).  Plain read-modify-write: every (n, k) of a problem has
// exactly one owner; problems that share a gradient buffer with another one (field 15 bit 30 set by the planner: the
// gradient-penalty terms of the encoder land on the discriminator's weights) use atomics.
// (at most 64 registers - it needed 65: what the phased NT kernel's two waves per SIMD leave free, so that this HBM-bound fold of one
//  branch runs beside the other branches' matrix launches instead of waiting for a CU to drain)
__global__ __launch_bounds__(256, 8) void tn_reduce_kernel(const int64_t* __restrict__ problems, const int32_t* __restrict__ red,
                                                        const float* __restrict__ ws, const float* __restrict__ alpha_dev) {
    const int32_t* r = red + 4 * blockIdx.x;
    const int pi = r[0], tile = r[1], first = r[2], splits = r[3], q = blockIdx.y;
    const int64_t* d = problems + 16 * pi;
    float* G = reinterpret_cast<float*>(d[4]);
    float* gbias = reinterpret_cast<float*>(d[5]);
    const int n_real = (int)d[10], k_real = (int)d[11], split_src = (int)d[12], gap = (int)d[13] - (int)d[12];
    const float alpha = __builtin_bit_cast(float, (int)d[14]) * (alpha_dev ? *alpha_dev : 1.f);
    const int tiles_k = (int)(d[15] & 0xFFFF), shared = (int)((d[15] >> 30) & 1);
    const int64_t stride = (int64_t)(((n_real + 255) / 256) * tiles_k) * kTnSlab;
    const int bn0 = (tile / tiles_k) * 256, bk0 = (tile % tiles_k) * 256;
    const float* base = ws + (int64_t)first * kTnSlab;
    // a workgroup owns 1024 consecutive 16-byte chunks of the tile image (gridDim.y = 16); a thread 4 of them, with the
    // loads of all four chunks (and of the gradient words they update) in flight together: the kernel is a pure stream of
    // (splits x 256 KB + 2 x gradient tile) per entry and must not serialise on one load latency per chunk
    int cidx[4], kk[4], n0[4];
    bool live[4];
    f32x4 sum[4];
    float gold[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = q * 1024 + u * 256 + threadIdx.x;
        const int lane = c & 63, cc = c >> 6, g = cc & 3, j = (cc >> 2) & 1, i = (cc >> 3) & 3, wid = cc >> 5;
        const int k = bk0 + ((wid & 3) * 2 + j) * 32 + (lane & 31);
        kk[u] = -1;
        if (k < split_src) kk[u] = k;
        else if (k >= split_src + gap && k - gap < k_real) kk[u] = k - gap;
        n0[u] = bn0 + ((wid >> 2) * 4 + i) * 32 + (lane >> 5) * 4 + 8 * g;
        cidx[u] = c;
        live[u] = kk[u] >= 0 && n0[u] < n_real;
        sum[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live[u]) {
            sum[u] = *reinterpret_cast<const f32x4*>(base + (int64_t)c * 4);
            if (!shared) {
#pragma unroll
                for (int e = 0; e < 4; ++e) gold[u][e] = (n0[u] + e < n_real) ? G[(int64_t)(n0[u] + e) * k_real + kk[u]] : 0.f;
            }
        }
    }
    for (int s2 = 1; s2 < splits; ++s2) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = live[u] ? *reinterpret_cast<const f32x4*>(base + s2 * stride + (int64_t)cidx[u] * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) sum[u][e] += v[u][e];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (!live[u]) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (n0[u] + e >= n_real) break;
            float* dst = G + (int64_t)(n0[u] + e) * k_real + kk[u];
            if (shared) atomic_add_f32(dst, alpha * sum[u][e]);
            else *dst = gold[u][e] + alpha * sum[u][e];
        }
    }
    if (q == 0 && gbias && bk0 == 0) {
        const int n = bn0 + threadIdx.x;
        if (n < n_real) {
            float t = 0.f;
            for (int s2 = 0; s2 < splits; ++s2) t += base[s2 * stride + 65536 + threadIdx.x];
            if (shared) atomic_add_f32(gbias + n, alpha * t);
            else gbias[n] += alpha * t;
        }
    }
}

template <typename T> int launch_tn8(TNParams p, hipStream_t stream) {
    constexpr int lds = 2 * 65536;
    static bool attr_done = false;
    auto kern = gemm_tn8_kernel<T>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            ase_set_error("gemm_tn8: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return ASE_ELAUNCH;
        }
        attr_done = true;
    }
    p.tiles_n = (p.n_real + 255) / 256;
    p.tiles_k = (p.K + 255) / 256;
    const int tiles = p.tiles_n * p.tiles_k;
    int splits = 256 / tiles;                                   // one 8-wave workgroup per CU
    const int max_splits = p.M / 256;                           // >= 4 K-tiles per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int chunk = (p.M + splits - 1) / splits;
    chunk = (chunk + 63) / 64 * 64;
    splits = (p.M + chunk - 1) / chunk;
    p.m_chunk = chunk;
    p.prof = g_nt_prof;
    ASE_LAUNCH(kern, dim3(tiles, 1, splits), dim3(512), lds, stream, p);
    ASE_CHECK_LAUNCH("gemm_tn8");
    return ASE_OK;
}

template <typename T> int launch_tn8g(const int64_t* problems, const int32_t* work, int n_work, const int32_t* red,
                                             int n_red, float* ws, const float* alpha_dev, hipStream_t stream) {
    constexpr int lds = 2 * 65536;
    static bool attr_done = false;
    auto kern = gemm_tn8g_kernel<T>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            ase_set_error("gemm_tn_grouped: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return ASE_ELAUNCH;
        }
        attr_done = true;
    }
    ASE_LAUNCH(kern, dim3(n_work), dim3(512), lds, stream, problems, work, n_work, g_nt_prof, ws, alpha_dev);
    if (ws) ASE_LAUNCH(tn_reduce_kernel, dim3(n_red, 16), dim3(256), 0, stream, problems, red, (const float*)ws, alpha_dev);
    ASE_CHECK_LAUNCH("gemm_tn_grouped");
    return ASE_OK;
}

template <typename T> int launch_tn(TNParams p, hipStream_t stream) {
    using Gm = TNGeom<T>;
    constexpr int lds = 4 * Gm::BKM * Gm::STRIDE;
    static bool attr_done = false;
    auto kern = gemm_tn_kernel<T>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            ase_set_error("gemm_tn: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return ASE_ELAUNCH;
        }
        attr_done = true;
    }
    p.tiles_n = (p.n_real + 127) / 128;
    p.tiles_k = (p.K + 127) / 128;
    const int tiles = p.tiles_n * p.tiles_k;
    // Split M so that the grid is ONE resident wave of workgroups: 80 KB of LDS => 2 workgroups per CU => 512 slots
    // on 256 CUs.  More splits only add f32 atomics (splits x N x K of them) and a partial second wave.
    constexpr int target_wg = 512;
    // narrow outputs (<= 8 tiles): the partial-sum atomics outweigh the second resident workgroup per CU (measured:
    // 256 workgroups beat 512 by 20-30 % on the head / style-MLP gradients)
    int splits = (tiles <= 8 ? 256 : target_wg) / tiles;
    const int max_splits = (p.M + 4 * Gm::BKM - 1) / (4 * Gm::BKM); // >= 4 staged tiles per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int chunk = (p.M + splits - 1) / splits;
    chunk = (chunk + Gm::BKM - 1) / Gm::BKM * Gm::BKM;
    splits = (p.M + chunk - 1) / chunk;
    p.m_chunk = chunk;
    ASE_LAUNCH(kern, dim3(tiles, 1, splits), dim3(kThreads), lds, stream, p);
    ASE_CHECK_LAUNCH("gemm_tn");
    return ASE_OK;
}

// ------------------------------------------------------------------------------------------------
// refresh_shadow: f32 master [n_real, k_real] -> dtype W_s [*, ldws] and transposed Wt_s [*, ldwts]
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void refresh_shadow_kernel(const float* __restrict__ W, int n_real, int k_real, T* __restrict__ Ws,
                                      int64_t ldws, T* __restrict__ Wts, int64_t ldwts, int split_src, int gap) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, k = k0 + tx;
        float v = 0.f;
        if (n < n_real && k < k_real) v = W[(int64_t)n * k_real + k];
        tile[ty + 8 * i][tx] = v;
        if (Ws && n < n_real && k < k_real) {
            const int kd = (k < split_src) ? k : k + gap;
            Ws[(int64_t)n * ldws + kd] = from_f32<T>(v);
        }
    }
    __syncthreads();
    if (Wts) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ty + 8 * i, n = n0 + tx;
            if (k < k_real && n < n_real) {
                const int kd = (k < split_src) ? k : k + gap;
                Wts[(int64_t)kd * ldwts + n] = from_f32<T>(tile[tx][ty + 8 * i]);
            }
        }
    }
}

// f32h_t shadows (ASE_F32H3): W * 2^e split into half hi / lo parts, packed [8 hi | 8 lo] per group of 8 consecutive elements of
// the contracted (contiguous) dimension - k for W_s, n for W_s^T; 4 bytes per element, leading dimensions in 4-byte units like the
// f32 shadows they replace.  (See Mma<f32h_t> in gemm_nt_kernels.h.)
__device__ __forceinline__ void store_split(char* row, int col, float v, float scale) {
    // (saturating, like every other conversion into half storage: a weight beyond +-32 at the scale 2^11 must not become inf and
    //  then NaN in every product - advisor, round 5; once per weight and step, free.  The A operand's split stays a plain cast in
    //  the kernel's loop: its inputs are bounded by construction - observations clamped to +-5 at 2^12, see _gp_value)
    const float s = __builtin_amdgcn_fmed3f(v * scale, -65504.f, 65504.f);
    const f16_t hi = (f16_t)s, lo = (f16_t)(s - (float)hi);
    char* g = row + (col >> 3) * 32 + (col & 7) * 2;
    *reinterpret_cast<f16_t*>(g) = hi;
    *reinterpret_cast<f16_t*>(g + 16) = lo;
}
__global__ void refresh_shadow_split_kernel(const float* __restrict__ W, int n_real, int k_real, char* __restrict__ Ws, int64_t ldws,
                                            char* __restrict__ Wts, int64_t ldwts, int split_src, int gap, float scale) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, k = k0 + tx;
        float v = 0.f;
        if (n < n_real && k < k_real) v = W[(int64_t)n * k_real + k];
        tile[ty + 8 * i][tx] = v;
        if (Ws && n < n_real && k < k_real) store_split(Ws + (int64_t)n * ldws * 4, (k < split_src) ? k : k + gap, v, scale);
    }
    __syncthreads();
    if (Wts) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ty + 8 * i, n = n0 + tx;
            if (k < k_real && n < n_real)
                store_split(Wts + (int64_t)((k < split_src) ? k : k + gap) * ldwts * 4, n, tile[tx][ty + 8 * i], scale);
        }
    }
}

// All layers in one launch: desc[l] = {W, n_real, k_real, Ws, ldws, Wts, ldwts, split_src, gap, bias, bias_shadow, tiles_k}
// (int64 each); blockIdx.y = layer, blockIdx.x = 32x32 tile (grid-stride), bias copied by the first workgroup.
template <typename T>
__global__ __launch_bounds__(256) void refresh_multi_kernel(const int64_t* __restrict__ desc) {
    __shared__ float tile[32][33];
    const int64_t* d = desc + 12 * blockIdx.y;
    const float* W = reinterpret_cast<const float*>(d[0]);
    const int n_real = (int)d[1], k_real = (int)d[2];
    T* Ws = reinterpret_cast<T*>(d[3]);
    const int64_t ldws = d[4];
    T* Wts = reinterpret_cast<T*>(d[5]);
    const int64_t ldwts = d[6];
    const int split_src = (int)d[7], gap = (int)d[8];
    const int tiles_k = (int)d[11];
    const int tiles = tiles_k * ((n_real + 31) / 32);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (blockIdx.x == 0 && d[9]) {
        const float* b = reinterpret_cast<const float*>(d[9]);
        float* bs = reinterpret_cast<float*>(d[10]);
        for (int i = threadIdx.x; i < n_real; i += 256) bs[i] = b[i];
    }
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int k0 = (t % tiles_k) * 32, n0 = (t / tiles_k) * 32;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + ty + 8 * i, k = k0 + tx;
            float v = 0.f;
            if (n < n_real && k < k_real) {
                v = W[(int64_t)n * k_real + k];
                Ws[(int64_t)n * ldws + ((k < split_src) ? k : k + gap)] = from_f32<T>(v);
            }
            tile[ty + 8 * i][tx] = v;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ty + 8 * i, n = n0 + tx;
            if (k < k_real && n < n_real)
                Wts[(int64_t)((k < split_src) ? k : k + gap) * ldwts + n] = from_f32<T>(tile[tx][ty + 8 * i]);
        }
    }
}

}  // namespace

extern "C" int ase_hip_refresh_shadow_multi(const int64_t* desc, int n_layers, int dtype, void* stream) {
    ASE_CHECK_ARG(desc && n_layers > 0, "refresh_shadow_multi: null/empty operand");
    const dim3 grid(256, n_layers);
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH(refresh_multi_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, desc);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "refresh_shadow_multi: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("refresh_shadow_multi");
    return ASE_OK;
}

extern "C" int ase_hip_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* G, float* gbias,
                               int bias_rows, int M, int N, int K, int n_real, int k_real, int split_src, int split_dst, float alpha,
                               const float* alpha_dev, int dtype, void* stream) {
    const int es = ase_elem_size(dtype);
    ASE_CHECK_ARG(dtype == ASE_F32 || dtype == ASE_BF16 || dtype == ASE_F32X3 || dtype == ASE_F16, "gemm_tn: bad dtype %d", dtype);
    ASE_CHECK_ARG(A && B && G && M > 0 && N > 0 && K > 0, "gemm_tn: null/empty operand");
    ASE_CHECK_ARG((N * es) % 16 == 0 && (K * es) % 16 == 0, "gemm_tn: N=%d / K=%d must cover whole 16-byte chunks", N, K);
    ASE_CHECK_ARG(lda >= N && ldb >= K, "gemm_tn: leading dimension too small");
    ASE_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && (lda * es) % 16 == 0 && (ldb * es) % 16 == 0,
                  "gemm_tn: A/B must be 16-byte aligned with 16-byte row pitch");
    ASE_CHECK_ARG(n_real > 0 && n_real <= N && k_real > 0 && split_src <= split_dst && split_src <= k_real,
                  "gemm_tn: bad real dims / split");
    TNParams p;
    p.A = (const char*)A; p.lda = lda * es; p.B = (const char*)B; p.ldb = ldb * es; p.G = G; p.gbias = gbias; p.bias_rows = bias_rows > 0 ? bias_rows : M;
    p.M = M; p.N = N; p.K = K; p.n_real = n_real; p.k_real = k_real; p.split_src = split_src; p.split_dst = split_dst;
    p.alpha = alpha; p.alpha_dev = alpha_dev; p.tiles_n = p.tiles_k = p.m_chunk = 0; p.prof = nullptr;
    if (es == 2) {
        // Single-problem launches take the phased 256 x 256 kernel only when few M-splits fill the chip (its split
        // reduction costs 256 KB of memory-side atomics per workgroup; see the grouped launch): whole 64-row K-tiles,
        // whole bias tiles, >= 32 K-tiles per split.
        const int t256 = ((n_real + 255) / 256) * ((K + 255) / 256);
        // (the phased kernel reaches its operands through 32-bit byte offsets: < 2 GiB each)
        const bool fits32 = (int64_t)M * p.lda < (int64_t)0x7FFFFFFF && (int64_t)M * p.ldb < (int64_t)0x7FFFFFFF;
        const bool phased = M % 64 == 0 && p.bias_rows % 64 == 0 && n_real >= 128 && K >= 128 && (int64_t)M * t256 >= 256 * 2048 && fits32;
        if (dtype == ASE_BF16) return phased ? launch_tn8<bf16_t>(p, (hipStream_t)stream) : launch_tn<bf16_t>(p, (hipStream_t)stream);
        return phased ? launch_tn8<f16_t>(p, (hipStream_t)stream) : launch_tn<f16_t>(p, (hipStream_t)stream);
    }
    if (dtype == ASE_F32X3) return launch_tn<f32s_t>(p, (hipStream_t)stream);
    return launch_tn<float>(p, (hipStream_t)stream);
}


// ---- grouped weight gradients (bf16) -----------------------------------------------------------------------------
static int tn_problem_check(const int64_t* d, int i) {
    const int64_t lda = d[1], ldb = d[3], bias_rows = d[6], M = d[7], N = d[8], K = d[9], n_real = d[10], k_real = d[11];
    ASE_CHECK_ARG(d[0] && d[2] && d[4] && M > 0 && N > 0 && K > 0, "gemm_tn_grouped: problem %d: null/empty operand", i);
    ASE_CHECK_ARG(M % 64 == 0 && (bias_rows <= 0 || bias_rows % 64 == 0),
                  "gemm_tn_grouped: problem %d: M=%lld / bias_rows=%lld must be whole 64-row K-tiles", i, (long long)M, (long long)bias_rows);
    ASE_CHECK_ARG(N % 8 == 0 && K % 8 == 0 && lda >= N && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 &&
                      ((uintptr_t)d[0] % 16) == 0 && ((uintptr_t)d[2] % 16) == 0,
                  "gemm_tn_grouped: problem %d: operands must be 16-byte aligned with whole 16-byte chunks per row", i);
    ASE_CHECK_ARG(n_real > 0 && n_real <= N && k_real > 0 && d[12] <= d[13] && d[12] <= k_real,
                  "gemm_tn_grouped: problem %d: bad real dims / split", i);
    ASE_CHECK_ARG(M * lda * 2 < (int64_t)0x7FFFFFFF && M * ldb * 2 < (int64_t)0x7FFFFFFF,
                  "gemm_tn_grouped: problem %d: operands beyond 2 GiB (the kernel addresses them with 32-bit byte offsets)", i);
    return ASE_OK;
}

extern "C" int ase_hip_gemm_tn_grouped_plan(int64_t* problems, int n_problems, int target_wg, int32_t* work, int max_work,
                                            int* n_work, int32_t* red, int max_red, int* n_red) {
    ASE_CHECK_ARG(problems && work && n_work && n_problems > 0 && max_work > 0, "gemm_tn_grouped_plan: null/empty argument");
    ASE_CHECK_ARG(red == nullptr || (n_red && max_red > 0), "gemm_tn_grouped_plan: reduce table without its size");
    if (target_wg <= 0) target_wg = 256;      // one 8-wave workgroup per CU
    int64_t max_kt = 1;
    for (int i = 0; i < n_problems; ++i) {
        int64_t* d = problems + 16 * i;
        const int rc = tn_problem_check(d, i);
        if (rc != ASE_OK) return rc;
        if (d[6] <= 0) d[6] = d[7];                          // bias_rows: all rows
        d[15] = (d[9] + 255) / 256;                          // tiles_k (bits 0-15)
        ASE_CHECK_ARG(d[15] < 65536, "gemm_tn_grouped_plan: problem %d: K too wide", i);
        for (int j = 0; j < n_problems; ++j)                 // bit 30: another problem adds to the same gradient / bias buffer
            if (j != i && (problems[16 * j + 4] == d[4] || (d[5] && problems[16 * j + 5] == d[5]))) d[15] |= (int64_t)1 << 30;
        if (d[7] / 64 > max_kt) max_kt = d[7] / 64;
    }
    // Contraction length c (K-tiles per work item): all tiles of a problem are cut at the same rows (workgroups on the
    // same rows of neighbouring tiles share operand panels in L2), one workgroup per CU is resident, and the grid runs
    // in ceil(items / target) rounds of ~c K-tiles each; every item also pays a prologue and 256 KB of atomics
    // (~8 K-tiles of main loop).  Pick the c with the shortest makespan.
    auto count = [&](int64_t c) {
        int64_t tot = 0;
        for (int i = 0; i < n_problems; ++i) {
            const int64_t* d = problems + 16 * i;
            const int64_t tiles = ((d[10] + 255) / 256) * (d[15] & 0xFFFF), kt = d[7] / 64;
            tot += tiles * ((kt + c - 1) / c);
        }
        return tot;
    };
    int64_t c = max_kt, best = -1;
    for (int64_t cc = (max_kt < 8 ? max_kt : 8); cc <= max_kt; ++cc) {
        const int64_t items = count(cc), rounds = (items + target_wg - 1) / target_wg;
        if (items > max_work) continue;
        const int64_t cost = rounds * (cc + 8);
        if (best < 0 || cost < best) { best = cost; c = cc; }
    }
    // Canonical numbering (the partial-tile slab an item writes; the reduce table refers to it): problem-major, then split,
    // then tile - split s of a tile sits `tiles of the problem` slabs further.
    struct Group { int prob, t0, nt, m_begin, nk, slab0; };
    std::vector<Group> groups;
    int nw = 0, nr = 0;
    for (int i = 0; i < n_problems; ++i) {
        const int64_t* d = problems + 16 * i;
        const int tiles = (int)(((d[10] + 255) / 256) * (d[15] & 0xFFFF));
        const int64_t kt = d[7] / 64, splits = (kt + c - 1) / c, chunk = (kt + splits - 1) / splits;
        int live = 0;                                        // splits that hold rows (the last ones may be empty)
        for (int64_t s = 0; s < splits; ++s) live += (s * chunk < kt);
        if (red) {
            for (int t = 0; t < tiles; ++t) {
                ASE_CHECK_ARG(nr < max_red, "gemm_tn_grouped_plan: more than %d reduce entries", max_red);
                red[4 * nr + 0] = i; red[4 * nr + 1] = t; red[4 * nr + 2] = nw + t; red[4 * nr + 3] = live;
                ++nr;
            }
        }
        for (int64_t s = 0; s < splits; ++s) {
            const int64_t k0 = s * chunk, nk = (k0 + chunk <= kt) ? chunk : kt - k0;
            if (nk <= 0) continue;
            ASE_CHECK_ARG(nk < 65536 && nw + tiles < 32768, "gemm_tn_grouped_plan: work item out of the packed range");
            groups.push_back(Group{i, 0, tiles, (int)(k0 * 64), (int)nk, nw});
            nw += tiles;
        }
    }
    ASE_CHECK_ARG(nw <= max_work, "gemm_tn_grouped_plan: more than %d work items", max_work);
    // Launch order.  Workgroup b runs on XCD b mod 8 and the kernel maps it to list position (b mod 8) * cap + b / 8, so
    // positions [x cap, (x + 1) cap) are XCD x's.  The tiles of one (problem, row range) read the same operand panels - 4 x 4
    // tiles of a 1024 x 1024 layer: 8 distinct panels for 32 panel reads - but only through ONE XCD's L2: a group that
    // straddles two XCDs is fetched twice.  So the groups are bin-packed (first fit, largest first) into the 8 position
    // ranges, whole, and the ranges are padded with empty items (nk = 0: the workgroup returns at once) to a common length.
    const int per_round = (target_wg + 7) / 8;
    const int rounds = (nw + target_wg - 1) / target_wg;
    int cap = (nw + 7) / 8;
    const int cap_max = (rounds * per_round > cap) ? rounds * per_round : cap;
    std::vector<Group> parts;                                // groups wider than a range: cut at multiples of cap_max
    for (const Group& g : groups)
        for (int t = 0; t < g.nt; t += cap_max)
            parts.push_back(Group{g.prob, g.t0 + t, (g.nt - t < cap_max) ? g.nt - t : cap_max, g.m_begin, g.nk, g.slab0});
    std::stable_sort(parts.begin(), parts.end(), [](const Group& a, const Group& b) { return a.nt > b.nt; });
    std::vector<int> bin_of(parts.size());
    for (;; ++cap) {
        int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool ok = true;
        for (size_t g = 0; g < parts.size() && ok; ++g) {
            int x = 0;
            while (x < 8 && fill[x] + parts[g].nt > cap) ++x;
            if (x == 8) { ok = false; break; }
            bin_of[g] = x;
            fill[x] += parts[g].nt;
        }
        if (ok) break;
        if (cap >= cap_max) {                                // no whole-group packing within the rounds: fill in order
            int x = 0, used = 0;
            std::vector<Group> cut;
            std::vector<int> cut_bin;
            for (const Group& g : parts) {
                int t = 0;
                while (t < g.nt) {
                    if (used == cap) { ++x; used = 0; }
                    const int n = (g.nt - t < cap - used) ? g.nt - t : cap - used;
                    cut.push_back(Group{g.prob, g.t0 + t, n, g.m_begin, g.nk, g.slab0});
                    cut_bin.push_back(x);
                    t += n; used += n;
                }
            }
            parts.swap(cut);
            bin_of.swap(cut_bin);
            break;
        }
    }
    ASE_CHECK_ARG(8 * cap <= max_work, "gemm_tn_grouped_plan: more than %d work items (%d with the XCD padding)", max_work, 8 * cap);
    for (int i = 0; i < 8 * cap * 4; ++i) work[i] = 0;
    int at[8];
    for (int x = 0; x < 8; ++x) at[x] = x * cap;
    for (size_t g = 0; g < parts.size(); ++g) {
        const Group& q = parts[g];
        for (int t = 0; t < q.nt; ++t) {
            int32_t* w = work + 4 * at[bin_of[g]]++;
            w[0] = q.prob; w[1] = q.t0 + t; w[2] = q.m_begin; w[3] = q.nk | ((q.slab0 + q.t0 + t) << 16);
        }
    }
    nw = 8 * cap;
    *n_work = nw;
    if (n_red) *n_red = nr;
    return ASE_OK;
}

extern "C" int ase_hip_gemm_tn_grouped(const int64_t* problems, const int32_t* work, int n_work, const int32_t* red, int n_red,
                                       float* workspace, const float* alpha_dev, int dtype, void* stream) {
    ASE_CHECK_ARG(problems && work && n_work > 0, "gemm_tn_grouped: null/empty argument");
    ASE_CHECK_ARG(dtype == ASE_BF16 || dtype == ASE_F16, "gemm_tn_grouped: 16-bit storage types only (dtype %d)", dtype);
    ASE_CHECK_ARG(workspace == nullptr || (red && n_red > 0 && ((uintptr_t)workspace % 16) == 0),
                  "gemm_tn_grouped: a workspace needs the reduce table of the plan (and 16-byte alignment)");
    if (dtype == ASE_F16) return launch_tn8g<f16_t>(problems, work, n_work, red, n_red, workspace, alpha_dev, (hipStream_t)stream);
    return launch_tn8g<bf16_t>(problems, work, n_work, red, n_red, workspace, alpha_dev, (hipStream_t)stream);
}

extern "C" int ase_hip_refresh_shadow(const float* W, int n_real, int k_real, void* Ws, int64_t ldws, void* Wts,
                                      int64_t ldwts, int split_src, int split_dst, int dtype, void* stream) {
    ASE_CHECK_ARG(W && n_real > 0 && k_real > 0 && (Ws || Wts), "refresh_shadow: null/empty operand");
    ASE_CHECK_ARG(split_src <= split_dst && split_src <= k_real, "refresh_shadow: bad split");
    const dim3 grid((k_real + 31) / 32, (n_real + 31) / 32), block(256);
    const int gap = split_dst - split_src;
    if ((dtype & 0xFF) == ASE_F32H3) {           // packed half split of W * 2^eb (eb in bits 16-23, as in ase_hip_gemm_nt's dtype word)
        const int eb = (dtype >> 16) & 0xFF;
        ASE_CHECK_ARG(eb <= 24 && ((dtype >> 8) & 0xFF) == 0, "refresh_shadow: ASE_F32H3 takes the weight scale exponent 0..24 in bits 16-23");
        ASE_CHECK_ARG((Ws == nullptr || (ldws % 8 == 0 && ((uintptr_t)Ws % 32) == 0)) && (Wts == nullptr || (ldwts % 8 == 0 && ((uintptr_t)Wts % 32) == 0)),
                      "refresh_shadow: packed split shadows need 32-byte aligned rows (leading dimensions in whole groups of 8)");
        ASE_LAUNCH(refresh_shadow_split_kernel, grid, block, 0, (hipStream_t)stream, W, n_real, k_real, (char*)Ws, ldws, (char*)Wts, ldwts,
                   split_src, gap, ldexpf(1.f, eb));
        ASE_CHECK_LAUNCH("refresh_shadow");
        return ASE_OK;
    }
    const int rc = ase_dispatch_storage(dtype, [&](auto tag) {
        typedef typename decltype(tag)::type T;
        ASE_LAUNCH(refresh_shadow_kernel<T>, grid, block, 0, (hipStream_t)stream, W, n_real, k_real, (T*)Ws, ldws, (T*)Wts, ldwts,
                   split_src, gap);
        return ASE_OK;
    });
    ASE_CHECK_ARG(rc == ASE_OK, "refresh_shadow: bad dtype %d", dtype);
    ASE_CHECK_LAUNCH("refresh_shadow");
    return ASE_OK;
}
