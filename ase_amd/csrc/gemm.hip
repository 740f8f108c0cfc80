// Matrix-core kernels for the dense layers of the ASE/AMP update (gfx950 / CDNA4).
//
//   gemm_nt : C = mask(act(alpha * A·Bᵀ + bias))      forward, data-gradient, gradient-penalty chain
//   gemm_tn : G += alpha * Aᵀ·B                        weight (+ bias) gradient, split over M, f32 atomics
//
// Storage types: bf16 (v_mfma_f32_32x32x16_bf16, f32 accumulate), exact f32 (v_mfma_f32_32x32x2_f32) and f32 multiplied
// as three bf16 MFMAs on a hi/lo split.  Wave64, XCD-aware tile order (8 XCDs, private L2s).
//   Staging: tiles go HBM -> LDS directly with global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass); LDS rows
//       are 128 bytes, unpadded (the DMA writes lane-linear), with the 16-byte chunks of row r stored at slot
//       chunk ^ ((r >> 1) & 7): the permutation is applied to the per-lane SOURCE address and again on the fragment
//       read, which makes every ds_read_b128 lane group hit 16 distinct bank slots.
//   NT kernels: gemm_nt_kernel (64 / 128 / 256 tiles, S-stage ring, one barrier per K-tile, every wave in lock-step) and
//       gemm_nt8_kernel (bf16, 256 x 256: four phases per K-tile, two wave groups one barrier apart, counted vmcnt) -
//       see the comment blocks in front of each; nt_choice() picks per shape.  One epilogue (nt_epilogue) for all.
//   TN kernels: gemm_tn_kernel (128 x 128, register-staged, transposed LDS reads) and the phased gemm_tn8 kernels
//       (256 x 256, DMA-staged), single problem or grouped (all weight gradients of a step in one grid).
#include "gemm_nt.h"
#include <stdlib.h>
#include <algorithm>
#include <vector>

using namespace ase_nt;

namespace {

unsigned long long* g_nt_prof = nullptr;      // tuning aid, see ase_hip_debug_nt_profile
int g_nt_prof_clk = 0;                        // ... stamps 1, 2 in shader clocks (ase_hip_debug_nt_profile_clock)

constexpr int kThreads = 256;

template <typename T> struct Mma;


template <typename T> struct Mma16 {
    typedef typename V16<T>::x8 x8;
    // one staged row = RB/2 k-values = RB/32 steps of 16
    // SW: D = B-fragment x A-fragment (transposed accumulator block: a lane owns one output row, see nt_epilogue_rows)
    template <int FM, int FN, int RB, bool SW = false>
    static __device__ __forceinline__ void tile(const char* sA, const char* sB, int lane, f32x16 (&acc)[FM][FN]) {
        const int r = lane & 31, h = lane >> 5, sw = lds_swz<RB>(r);
#pragma unroll
        for (int ks = 0; ks < RB / 32; ++ks) {
            x8 a[FM], b[FN];
            const int off = r * RB + (((ks * 2 + h) ^ sw) << 4);
#pragma unroll
            for (int i = 0; i < FM; ++i)
                a[i] = *reinterpret_cast<const x8*>(sA + i * 32 * RB + off);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                b[j] = *reinterpret_cast<const x8*>(sB + j * 32 * RB + off);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if constexpr (SW) acc[i][j] = mfma16<T>(b[j], a[i], acc[i][j]);
                    else acc[i][j] = mfma16<T>(a[i], b[j], acc[i][j]);
                }
        }
    }
};
template <> struct Mma<bf16_t> : Mma16<bf16_t> {};
template <> struct Mma<f16_t> : Mma16<f16_t> {};

template <> struct Mma<float> {
    // one staged row = RB/4 k-values = RB/32 blocks of 8; within a block lane-half h holds k = 4h..4h+3
    // and MFMA j multiplies element j of both operands (any k order is fine if A and B agree).
    template <int FM, int FN, int RB>
    static __device__ __forceinline__ void tile(const char* sA, const char* sB, int lane, f32x16 (&acc)[FM][FN]) {
        const int r = lane & 31, h = lane >> 5, sw = lds_swz<RB>(r);
#pragma unroll
        for (int kb = 0; kb < RB / 32; ++kb) {
            f32x4 a[FM], b[FN];
            const int off = r * RB + (((kb * 2 + h) ^ sw) << 4);
#pragma unroll
            for (int i = 0; i < FM; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(sA + i * 32 * RB + off);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(sB + j * 32 * RB + off);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    }
};

__device__ __forceinline__ void split_bf16(const f32x4& x0, const f32x4& x1, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        hi[q] = (bf16_t)x0[q];
        hi[q + 4] = (bf16_t)x1[q];
        lo[q] = (bf16_t)(x0[q] - (float)hi[q]);
        lo[q + 4] = (bf16_t)(x1[q] - (float)hi[q + 4]);
    }
}

template <> struct Mma<f32s_t> {
    // f32 rows (RB/4 k-values = RB/64 steps of 16): a lane needs 8 consecutive k per step = 2 chunks.
    template <int FM, int FN, int RB>
    static __device__ __forceinline__ void tile(const char* sA, const char* sB, int lane, f32x16 (&acc)[FM][FN]) {
        const int r = lane & 31, h = lane >> 5, sw = lds_swz<RB>(r);
#pragma unroll
        for (int ks = 0; ks < RB / 64; ++ks) {
            bf16x8 ah[FM], al[FM], bh[FN], bl[FN];
            const int c0 = ks * 4 + h * 2;
            const int o0 = r * RB + ((c0 ^ sw) << 4), o1 = r * RB + (((c0 + 1) ^ sw) << 4);
#pragma unroll
            for (int i = 0; i < FM; ++i)
                split_bf16(*reinterpret_cast<const f32x4*>(sA + i * 32 * RB + o0),
                           *reinterpret_cast<const f32x4*>(sA + i * 32 * RB + o1), ah[i], al[i]);
#pragma unroll
            for (int j = 0; j < FN; ++j)
                split_bf16(*reinterpret_cast<const f32x4*>(sB + j * 32 * RB + o0),
                           *reinterpret_cast<const f32x4*>(sB + j * 32 * RB + o1), bh[j], bl[j]);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                }
        }
    }
};

// HBM -> LDS DMA of one operand tile: pass i moves rows [RPP i, RPP i + RPP) (RPP = 8 rows per wave); wave w of the
// pass owns the 8 rows RPP i + 8 w .. + 7 = one 1-KiB lane-linear LDS piece (M0 = wave-uniform base, lane l lands at
// base + 16 l).
template <int PASSES, int RPP, int RB>
__device__ __forceinline__ void nt_stage(const char* const (&src)[PASSES], int64_t koff, char* lds_wave_base) {
#pragma unroll
    for (int i = 0; i < PASSES; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t*)(src[i] + koff), (lptr_t*)(lds_wave_base + i * RPP * RB), 16, 0, 0);
}
// ---- epilogue of the NT kernels.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (e&3) + 8*(e>>2) + 4*(lane>>5),
// i.e. a lane owns ONE column: storing from registers would be 2-byte scattered stores.  Phase 1 applies bias +
// activation (per-column bias = per-lane scalar) and transposes FMC x FNC fragments of the wave's sub-tile through a
// wave-private f32 LDS slab [FMC*32][FNC*32] (row pitch 64 dwords: ds_write_b32 and ds_read_b128 are both
// conflict-free); phase 2 lets every lane pick up 4 consecutive columns of a row, applies the derivative mask,
// issues 8/16-byte row-contiguous stores (full 128-byte lines per row) and keeps per-column partial sums for the bias
// gradient.  The caller guarantees that nobody still reads the staging ring (barrier).
//   The mask operand (AUXK = 1: the activation itself, 8/16 bytes per lane and row; AUXK = 2: its bit matrix, one word)
// is loaded a whole chunk AHEAD of its use - all rows of a chunk at once, the next chunk's before the current chunk's
// LDS transposition: left inside the row loop the loads cost one exposed HBM round trip per 4 rows (+10 us on a
// 256 x 256 tile, measured, whatever their width).
template <typename T, int AUXK> struct AuxReg;
template <typename T> struct AuxReg<T, 0> { char v; };
template <> struct AuxReg<bf16_t, 1> { bf16x4 v; };
template <> struct AuxReg<f16_t, 1> { f16x4 v; };
template <> struct AuxReg<float, 1> { f32x4 v; };
template <> struct AuxReg<f32s_t, 1> { f32x4 v; };
template <typename T> struct AuxReg<T, 2> { uint32_t v; };

// OR over aligned groups of 8 lanes with DPP only (no LDS round trip): xor 1, xor 2, then the half-row mirror
__device__ __forceinline__ uint32_t or8_dpp(uint32_t x) {
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, true);   // row_half_mirror
    return x;
}

// mask operand of chunk ch (jc-major) of a wave's sub-tile: one AuxReg per row iteration
template <typename T, int FM, int FN, int FMC, int FNC, int AUXK>
__device__ __forceinline__ void nt_aux_load(const NTParams& p, int ch, int lane, int mrow0, int ncol0,
                                            AuxReg<T, AUXK> (&dst)[(FMC * 32) / (64 / (FNC * 8))]) {
    constexpr int ELPR = FNC * 8, RPI = 64 / ELPR, NIT = (FMC * 32) / RPI;
    if constexpr (AUXK != 0) {
        const int c4 = lane % ELPR, rsub = lane / ELPR;
        const int jc = (ch / (FM / FMC)) * FNC, ic = (ch % (FM / FMC)) * FMC;
        const int n0 = ncol0 + jc * 32 + c4 * 4;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int m = mrow0 + ic * 32 + it * RPI + rsub;
            if (n0 < p.N && m < p.M) {
                const int ma = (m >= p.aux_split) ? m - p.aux_delta : m;
                if constexpr (AUXK == 2)
                    dst[it].v = *reinterpret_cast<const uint32_t*>(p.aux + (int64_t)ma * p.ldaux + (n0 >> 5) * 4);
                else
                    dst[it].v = *reinterpret_cast<const decltype(dst[it].v)*>(p.aux + (int64_t)ma * p.ldaux + (int64_t)n0 * sizeof(T));
            }
        }
    }
}

// PRE: chunk 0 of the mask operand was loaded by the caller (before its main loop) into pre[]
template <typename T, int FM, int FN, int FMC, int FNC, int AUXK, bool PRE = false>
__device__ __forceinline__ void nt_epilogue_impl(const NTParams& p, f32x16 (&acc)[FM][FN], float* slab, int lane,
                                                 int mrow0, int ncol0,
                                                 AuxReg<T, AUXK> (*pre)[(FMC * 32) / (64 / (FNC * 8))] = nullptr) {
    constexpr int WCOLS = FNC * 32, WROWS = FMC * 32;
    const int col_in = lane & 31, row_hi = (lane >> 5) * 4;
    constexpr int ELPR = WCOLS / 4;                // lanes per row (4 columns each)
    constexpr int RPI = 64 / ELPR;                 // rows per iteration
    constexpr int NIT = WROWS / RPI;               // row iterations per chunk
    constexpr int NCH = (FN / FNC) * (FM / FMC);   // chunks, jc-major
    constexpr bool AHEAD = AUXK != 0 && sizeof(AuxReg<T, AUXK>) <= 8;   // 16-byte f32 masks: current chunk only
    const int c4 = lane % ELPR, rsub = lane / ELPR;
    AuxReg<T, AUXK> areg[AHEAD ? 2 : 1][NIT];

    auto load_aux = [&](int ch, AuxReg<T, AUXK> (&dst)[NIT]) {
        nt_aux_load<T, FM, FN, FMC, FNC, AUXK>(p, ch, lane, mrow0, ncol0, dst);
    };

    if constexpr (PRE) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) areg[0][it] = (*pre)[it];
    } else if constexpr (AUXK != 0) {
        load_aux(0, areg[0]);
    }
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        const int jc = (ch / (FM / FMC)) * FNC, ic = (ch % (FM / FMC)) * FMC;
        const int n0 = ncol0 + jc * 32 + c4 * 4;
        if constexpr (AHEAD) {
            if (ch + 1 < NCH) load_aux(ch + 1, areg[(ch + 1) & 1]);
        } else if constexpr (AUXK != 0) {
            if (ch > 0) load_aux(ch, areg[0]);
        }
        AuxReg<T, AUXK> (&cur)[NIT] = areg[AHEAD ? (ch & 1) : 0];
#pragma unroll
        for (int jj = 0; jj < FNC; ++jj) {
            const int j = jc + jj;
            const int n = ncol0 + j * 32 + col_in;
            const float bias = (p.bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
            for (int ii = 0; ii < FMC; ++ii) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = p.alpha * acc[ic + ii][j][e] + bias;
                    if (p.act == ASE_ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.act == ASE_ACT_TANH) v = tanhf(v);
                    slab[(ii * 32 + row_hi + (e & 3) + 8 * (e >> 2)) * WCOLS + jj * 32 + col_in] = v;
                }
            }
        }
        float cs[4] = {0.f, 0.f, 0.f, 0.f};
        if (n0 < p.N) {                                // N is a multiple of 4 (checked on the host)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int row = it * RPI + rsub;
                const int m = mrow0 + ic * 32 + row;
                if (m >= p.M) continue;
                f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * WCOLS + c4 * 4);
                if (p.act >= ASE_ACT_SILU) {            // smooth activations: the slab holds z; keep it (twin), then activate
                    if (p.pre_out) {
                        if constexpr (sizeof(T) == 2) {
                            typename V16<T>::x4 zt;
#pragma unroll
                            for (int q = 0; q < 4; ++q) zt[q] = from_f32<T>(v[q]);
                            *reinterpret_cast<typename V16<T>::x4*>(p.pre_out + (int64_t)m * p.ldpre + (int64_t)n0 * 2) = zt;
                        } else {
                            *reinterpret_cast<f32x4*>(p.pre_out + (int64_t)m * p.ldpre + (int64_t)n0 * 4) = v;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = act_apply(p.act, v[q]);
                }
                if constexpr (AUXK == 2) {
                    const uint32_t nib = cur[it].v >> (n0 & 31);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = ((nib >> q) & 1u) ? v[q] : 0.f;
                } else if constexpr (AUXK == 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float a = (float)cur[it].v[q];
                        if (p.aux_mode == ASE_AUX_RELU_MASK) v[q] = a > 0.f ? v[q] : 0.f;
                        else if (p.aux_mode == ASE_AUX_TANH_GRAD) v[q] = v[q] * (1.f - a * a);
                        else v[q] = v[q] * act_grad(p.aux_mode >> 8, a);          // ASE_AUX_PREACT | (activation << 8)
                    }
                }
                if (p.out_f32 || sizeof(T) == 4) {
                    *reinterpret_cast<f32x4*>(p.C + (int64_t)m * p.ldc + (int64_t)n0 * 4) = v;
                } else {
                    typedef typename std::conditional<sizeof(T) == 2, T, bf16_t>::type S;     // (4-byte T: dead branch)
                    typename V16<S>::x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[q] = from_f32<S>(v[q]);
                        v[q] = (float)o[q];
                    }
                    *reinterpret_cast<typename V16<S>::x4*>(p.C + (int64_t)m * p.ldc + (int64_t)n0 * 2) = o;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) cs[q] += v[q];
                if (p.mask_out) {
                    // nibble of this lane's 4 columns -> OR over the 8 lanes of a 32-column word -> one 4-byte store
                    uint32_t bits = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) bits |= (v[q] > 0.f ? 1u : 0u) << q;
                    bits = or8_dpp(bits << (n0 & 31));
                    if ((c4 & 7) == 0) p.mask_out[(int64_t)m * p.ldmask + (n0 >> 5)] = bits;
                }
            }
        }
        // bias gradient: the chunks of one column group (same jc) follow each other; flush after the last of them
        if (p.colsum) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int o = ELPR; o < 64; o <<= 1) cs[q] += __shfl_xor(cs[q], o, 64);
            }
            if (lane < ELPR) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n0 + q < p.colsum_n) atomic_add_f32(p.colsum + n0 + q, cs[q]);
            }
        }
    }
}

template <typename T, int FM, int FN, int FMC, int FNC>
__device__ __forceinline__ void nt_epilogue(const NTParams& p, f32x16 (&acc)[FM][FN], float* slab, int lane, int mrow0,
                                            int ncol0) {
    if (p.aux_mode == ASE_AUX_NONE) nt_epilogue_impl<T, FM, FN, FMC, FNC, 0>(p, acc, slab, lane, mrow0, ncol0);
    else if (p.aux_mode == ASE_AUX_RELU_BITS) nt_epilogue_impl<T, FM, FN, FMC, FNC, 2>(p, acc, slab, lane, mrow0, ncol0);
    else nt_epilogue_impl<T, FM, FN, FMC, FNC, 1>(p, acc, slab, lane, mrow0, ncol0);
}

// ---- row-per-lane epilogue of the lock-step kernels (bf16, swapped MFMA operands): the general FM x FN form of
// nt8_epilogue_rows further down - see there.  acc[i][j]: lane (r = lane & 31, h = lane >> 5) owns output row i*32 + r
// and the columns j*32 + 8 g + 4 h + q.  bits[i][j]: the ReLU mask word of (row, 32-column fragment), loaded by the caller.
template <typename T, int FM, int FN, int AUXK>
__device__ __forceinline__ void nt_epilogue_rows(const NTParams& p, f32x16 (&acc)[FM][FN], int lane, int mrow0, int ncol0,
                                                 const uint32_t (&bits)[FM][FN]) {
    if (ncol0 >= p.N) return;                                   // wave-uniform: N is a multiple of the wave tile's width
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
        f32x4 bias[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (p.bias) bias[g] = *reinterpret_cast<const f32x4*>(p.bias + ncol0 + j * 32 + 8 * g + 4 * h);
            else bias[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int m = mrow0 + i * 32 + r;
            const bool row_ok = m < p.M;
            char* crow = p.C + (int64_t)m * p.ldc + (int64_t)(ncol0 + j * 32) * 2 + h * 16;
            uint32_t pk[4][2];
            uint32_t mb = 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                T o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = p.alpha * acc[i][j][g * 4 + q] + bias[g][q];
                    if (p.act == ASE_ACT_RELU) v = fmaxf(v, 0.f);
                    if constexpr (AUXK == 2) v = ((bits[i][j] >> (8 * g + 4 * h + q)) & 1u) ? v : 0.f;
                    o[q] = from_f32<T>(v);
                    mb |= ((float)o[q] > 0.f ? 1u : 0u) << (8 * g + 4 * h + q);
                }
                pk[g][0] = (uint32_t)__builtin_bit_cast(uint16_t, o[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, o[1]) << 16);
                pk[g][1] = (uint32_t)__builtin_bit_cast(uint16_t, o[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, o[3]) << 16);
            }
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                if (row_ok) *reinterpret_cast<uint4*>(crow + 8 * g * 2) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            }
            if (p.mask_out) {
                const auto w = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);   // own 16 bits | the other half-wave's
                if (row_ok && h == 0) p.mask_out[(int64_t)m * p.ldmask + ((ncol0 + j * 32) >> 5)] = w[0] | w[1];
            }
        }
    }
}

// WPE = minimum waves per SIMD the register allocation must leave room for (k workgroups of T threads per CU <=> k T / 256)
// SW (bf16): swapped MFMA operands + row-per-lane epilogue (16-byte stores from registers, no LDS slab)
template <typename T, int WGM, int WGN, int FM, int FN, int RB, int S, int WPE = 1, bool SW = false>
__global__ __launch_bounds__(WGM * WGN * 64, WPE) void gemm_nt_kernel(NTParams p) {
    constexpr int BM = WGM * FM * 32, BN = WGN * FN * 32;
    constexpr int BK = RB / (int)sizeof(T);
    constexpr int LPR = RB / 16;                             // lanes (16-byte chunks) per staged row
    constexpr int RPW = 64 / LPR;                            // rows per wave-instruction of the DMA
    constexpr int RPP = WGM * WGN * RPW;                     // tile rows staged per pass
    constexpr int A_PASSES = BM / RPP, B_PASSES = BN / RPP;
    constexpr int P = A_PASSES + B_PASSES;                   // DMA instructions per lane per K-tile
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the rows staged per pass");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kBuf = (BM + BN) * RB;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WGN, wn = wid % WGN;
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int bm0 = (tile / p.tiles_n) * BM, bn0 = (tile % p.tiles_n) * BN;

    // per-lane DMA sources: tile row RPP i + tid / LPR, LDS slot tid % LPR receives chunk slot ^ swz(row).
    // Rows past M / N are clamped to the last valid row: their products only reach output rows / columns that are
    // never stored.
    const int srow = tid / LPR, sslot = tid % LPR;
    const char* srcA[A_PASSES];
    const char* srcB[B_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int r = i * RPP + srow;
        srcA[i] = p.A + (int64_t)min(bm0 + r, p.M - 1) * p.lda + ((sslot ^ lds_swz<RB>(r)) << 4);
    }
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
        const int r = i * RPP + srow;
        srcB[i] = p.B + (int64_t)min(bn0 + r, p.N - 1) * p.ldb + ((sslot ^ lds_swz<RB>(r)) << 4);
    }
    char* const ldsA = smem + (wid * RPW) * RB;
    char* const ldsB = ldsA + BM * RB;

    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // row-per-lane epilogue: the wave tile's mask words.  With a 2-stage ring every K-tile waits vmcnt(0), so the words
    // can be requested up front (they retire with the first K-tile wherever the compiler places the loads); deeper rings
    // use counted waits and fetch them after the loop.
    uint32_t row_bits[FM][FN];
    auto load_bits = [&]() {
        if (p.aux_mode == ASE_AUX_RELU_BITS && bn0 + wn * FN * 32 < p.N) {
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = min(bm0 + wm * FM * 32 + i * 32 + (lane & 31), p.M - 1);
                const int ma = (m >= p.aux_split) ? m - p.aux_delta : m;
                const uint32_t* w = reinterpret_cast<const uint32_t*>(p.aux + (int64_t)ma * p.ldaux) + ((bn0 + wn * FN * 32) >> 5);
#pragma unroll
                for (int j = 0; j < FN; ++j) row_bits[i][j] = w[j];
            }
        }
    };
    if constexpr (SW && S == 2) load_bits();

    // S-stage ring of LDS buffers, DMA prefetch distance S-1 tiles, ONE barrier per K-tile:
    //   wait (counted vmcnt: only the newest S-2 tiles may still be in flight) -> barrier (tile kt has landed for
    //   every wave AND every wave is done reading tile kt-1) -> issue the DMA of tile kt+S-1 into the buffer tile
    //   kt-1 occupied -> MFMAs on tile kt.
    const int nk = p.K / BK;
#pragma unroll
    for (int t = 0; t < S - 1; ++t) {
        if (t < nk) {
            nt_stage<A_PASSES, RPP, RB>(srcA, (int64_t)t * RB, ldsA + t * kBuf);
            nt_stage<B_PASSES, RPP, RB>(srcB, (int64_t)t * RB, ldsB + t * kBuf);
        }
    }
    int buf = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + S - 2 < nk) wait_vmcnt<P*(S - 2)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + S - 1 < nk) {
            const int nb = (buf == 0) ? S - 1 : buf - 1;           // (kt + S - 1) % S
            nt_stage<A_PASSES, RPP, RB>(srcA, (int64_t)(kt + S - 1) * RB, ldsA + nb * kBuf);
            nt_stage<B_PASSES, RPP, RB>(srcB, (int64_t)(kt + S - 1) * RB, ldsB + nb * kBuf);
        }
        const char* sA = smem + buf * kBuf + (wm * FM * 32) * RB;
        const char* sB = smem + buf * kBuf + (BM + wn * FN * 32) * RB;
        if constexpr (SW && sizeof(T) == 2) Mma<T>::template tile<FM, FN, RB, true>(sA, sB, lane, acc);
        else Mma<T>::template tile<FM, FN, RB>(sA, sB, lane, acc);
        buf = (buf + 1 == S) ? 0 : buf + 1;
    }
    if constexpr (SW) {
        if constexpr (S != 2) load_bits();
        if constexpr (sizeof(T) == 2) {
            if (p.aux_mode == ASE_AUX_RELU_BITS) nt_epilogue_rows<T, FM, FN, 2>(p, acc, lane, bm0 + wm * FM * 32, bn0 + wn * FN * 32, row_bits);
            else nt_epilogue_rows<T, FM, FN, 0>(p, acc, lane, bm0 + wm * FM * 32, bn0 + wn * FN * 32, row_bits);
        }
        return;
    }
    __syncthreads();                                   // everyone is done with the ring before it becomes the epilogue slab

    constexpr int FNC = (FN > 2) ? 2 : FN;
    float* slab = reinterpret_cast<float*>(smem) + wid * (FM * 32 * FNC * 32);
    nt_epilogue<T, FM, FN, FM, FNC>(p, acc, slab, lane, bm0 + wm * FM * 32, bn0 + wn * FN * 32);
}

template <typename T, int WGM, int WGN, int FM, int FN, int RB, int S, int WPE = 1, bool SW = false>
int launch_nt(const NTParams& p0, hipStream_t stream) {
    constexpr int BM = WGM * FM * 32, BN = WGN * FN * 32;
    constexpr int ring = S * (BM + BN) * RB;
    constexpr int slab = WGM * WGN * (FM * 32) * ((FN > 2 ? 2 : FN) * 32) * 4;
    constexpr int lds = ring > slab ? ring : slab;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static bool attr_done = false;
    auto kern = gemm_nt_kernel<T, WGM, WGN, FM, FN, RB, S, WPE, SW>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            ase_set_error("gemm_nt: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return ASE_ELAUNCH;
        }
        attr_done = true;
    }
    NTParams p = p0;
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    ASE_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n), dim3(WGM * WGN * 64), lds, stream, p);
    ASE_CHECK_LAUNCH("gemm_nt");
    return ASE_OK;
}

// ------------------------------------------------------------------------------------------------
// NT, phased 256 x 256 kernel (bf16).  512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 outputs = four
// 64 x 32 quadrants.  One K-tile (64 k-values, 128-byte rows, the swizzle of the kernel above) is FOUR phases, each
//     ds_read the fragments of one quadrant | issue one 16-KiB DMA unit (2 global_load_lds per lane) | counted vmcnt
//     s_barrier | lgkmcnt(0) | 8 MFMAs 32x32x16 | s_barrier
// and the two wave groups (waves 0-3 / 4-7: the two waves of every SIMD sit in different groups) run ONE BARRIER
// APART, so that on each SIMD one wave is in its MFMA block while its partner reads LDS and issues the DMA.
//   DMA units of K-tile t, in issue order = order of first use:
//     A0 = A rows {0-63, 128-191} (sub-tile 0 of both wave rows)     read in phase 0
//     B0 = B rows {64 c .. 64 c + 31, c = 0..3} (fragment 0 of every wave column)   phase 0 (kept in registers to phase 3)
//     B1 = B rows {64 c + 32 .. 64 c + 63}                            phase 1
//     A1 = A rows {64-127, 192-255}                                   phase 2
//   unit u = 4 t + kind is issued in phase (t', p) with 4 t' + p + 6 = u: six units ahead, into the buffer (t & 1) whose
//   previous occupant (K-tile t - 2) was last read >= 2 phases earlier (the WAR distance two staggered groups need);
//   a unit is read one phase after the counted wait + barrier that retires it (RAW across the stagger).
// ------------------------------------------------------------------------------------------------

template <int UNITS> __device__ __forceinline__ void wait_dma_units() { wait_vmcnt<2 * UNITS>(); }
__device__ __forceinline__ void wait_dma_units_rt(int units) {      // wave-uniform runtime count (loop tail)
    if (units >= 4) wait_vmcnt<8>();
    else if (units == 3) wait_vmcnt<6>();
    else if (units == 2) wait_vmcnt<4>();
    else if (units == 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
}

struct NT8Lane {
    const char* src[4][2];     // per-lane DMA source (row base + swizzled chunk) of unit kind x piece
    int dst[4][2];             // wave-uniform LDS byte offset of the piece inside a K-tile buffer
    int roff[4];               // per-lane fragment read offsets (row * 128 + swizzled chunk) for the 4 k-steps
};

template <int KIND, bool LIVE = true>
__device__ __forceinline__ void nt8_issue(const NT8Lane& L, char* smem, int tile) {
    if constexpr (!LIVE) return;
    constexpr int kBuf = 512 * 128;
    char* buf = smem + (tile & 1) * kBuf;
    const int64_t koff = (int64_t)tile * 128;
#pragma unroll
    for (int g = 0; g < 2; ++g)
        __builtin_amdgcn_global_load_lds((gptr_t*)(L.src[KIND][g] + koff), (lptr_t*)(buf + L.dst[KIND][g]), 16, 0, 0);
}

// fragment registers of one 32-row operand block: 4 k-steps x 16 bytes
template <bool LIVE = true>
__device__ __forceinline__ void nt8_read(i32x4 (&f)[4], const char* base, const NT8Lane& L) {
    if constexpr (!LIVE) { asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); return; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const i32x4*>(base + L.roff[ks]);
}

template <typename T, bool LIVE = true, bool SW = false>
__device__ __forceinline__ void nt8_mma(f32x16& c0, f32x16& c1, const i32x4 (&a0)[4], const i32x4 (&a1)[4],
                                        const i32x4 (&b)[4]) {
    if constexpr (!LIVE) { asm volatile("" : "+v"(c0), "+v"(c1) : "v"(a0[0]), "v"(a1[3]), "v"(b[2])); return; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        c0 = nt8_mfma<T, SW>(a0[ks], b[ks], c0);
        c1 = nt8_mfma<T, SW>(a1[ks], b[ks], c1);
    }
}

// the same 8 MFMAs with the two DMA pieces of unit KIND (K-tile `tile`) issued among them: an LDS-DMA instruction costs
// ~60 issue cycles beside MFMAs (the matrix pipe stays fed by the 32-cycle MFMA issue cadence) but 100-185 cycles in the
// read half of a phase, where it sat on the critical path of the OTHER wave group's MFMA block (measured by ablation:
// DMA and fragment reads were additive on top of the MFMA time)
template <typename T, int KIND, bool DM, bool MM, bool SW, bool MS = false>
__device__ __forceinline__ void nt8_mma_issue(f32x16& c0, f32x16& c1, const i32x4 (&a0)[4], const i32x4 (&a1)[4],
                                              const i32x4 (&b)[4], const NT8Lane& L, char* smem, int tile, bool live) {
    if constexpr (!MM) {
        asm volatile("" : "+v"(c0), "+v"(c1) : "v"(a0[0]), "v"(a1[3]), "v"(b[2]));
        if (DM && live) nt8_issue<KIND>(L, smem, tile);
        return;
    }
    constexpr int kBuf = 512 * 128;
    char* buf = smem + (tile & 1) * kBuf;
    const int64_t koff = (int64_t)tile * 128;
#ifdef ASE_LAB
    if constexpr (MS && std::is_same<T, bf16_t>::value) {
        // Lab ablation (TIMING ONLY, wrong results): the phase's 8 x 32x32x16 MFMAs on two accumulators replaced by the same
        // flop count as 16 x 16x16x32 MFMAs on EIGHT independent 4-register accumulators (the pieces of c0 / c1), each used
        // twice eight issues apart - the instruction mix of the 16x16x32 form of this schedule.  Question it answers: is the
        // loop's ~80 % matrix-pipe occupancy a property of the 32x32x16 issue / dependency cadence?
        f32x4 q[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            q[e] = f32x4{c0[4 * e], c0[4 * e + 1], c0[4 * e + 2], c0[4 * e + 3]};
            q[4 + e] = f32x4{c1[4 * e], c1[4 * e + 1], c1[4 * e + 2], c1[4 * e + 3]};
        }
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const int ks = (n >> 1) & 3;
            const bf16x8 av = __builtin_bit_cast(bf16x8, (n & 1) ? a1[ks] : a0[ks]), bv = __builtin_bit_cast(bf16x8, b[ks]);
            q[n & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bv, av, q[n & 7], 0, 0, 0);
            if (DM && (n == 3 || n == 11)) {
                __builtin_amdgcn_sched_barrier(0);
                if (live)
                    __builtin_amdgcn_global_load_lds((gptr_t*)(L.src[KIND][n >> 3] + koff), (lptr_t*)(buf + L.dst[KIND][n >> 3]), 16, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < 4; ++t) { c0[4 * e + t] = q[e][t]; c1[4 * e + t] = q[4 + e][t]; }
        return;
    }
#endif
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        c0 = nt8_mfma<T, SW>(a0[ks], b[ks], c0);
        c1 = nt8_mfma<T, SW>(a1[ks], b[ks], c1);
        if (DM && (ks == 0 || ks == 2)) {
            __builtin_amdgcn_sched_barrier(0);
            if (live)
                __builtin_amdgcn_global_load_lds((gptr_t*)(L.src[KIND][ks >> 1] + koff), (lptr_t*)(buf + L.dst[KIND][ks >> 1]), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// schedule variants (bit mask V): 1 = retire the LDS reads BEFORE the first barrier; 2 = no s_setprio around the MFMAs.
// (The DMA is always issued AFTER the phase's fragment reads: hipcc puts a vmcnt(0) in front of any LDS read that
// follows a global_load_lds without a barrier in between.)
template <int V> __device__ __forceinline__ void nt8_sync_in() {      // end of the read / issue half of a phase
    if constexpr (V & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    NT8_BARRIER();
    if constexpr (!(V & 1)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (!(V & 2)) __builtin_amdgcn_s_setprio(1);
}
template <int V> __device__ __forceinline__ void nt8_sync_out() {     // end of the MFMA half
    if constexpr (!(V & 2)) __builtin_amdgcn_s_setprio(0);
    NT8_BARRIER();
}

// one K-tile = 4 phases.  TAIL = false: every issued unit exists (t + 2 < nk) and the waits are compile-time counts.
// V & 64: the DMA of a phase is issued INSIDE its MFMA block (nt8_mma_issue).  Unit 4 t + p + 6 still belongs to phase
// (t, p), but at the counted wait of a phase (in front of its first barrier) the newest issued unit is now the one of the
// previous phase: three units may stay in flight instead of four.  RAW (read one phase after wait + barrier) is unchanged,
// the WAR distance grows by half a phase.
template <typename T, bool TAIL, int V, bool SW>
__device__ __forceinline__ void nt8_ktile_m(int t, int nk, const NT8Lane& L, char* smem, const char* aP, const char* bP,
                                            f32x16 (&acc)[4][2], i32x4 (&a0)[4], i32x4 (&a1)[4], i32x4 (&b0)[4],
                                            i32x4 (&b1)[4]) {
    constexpr int RB = 128;
    constexpr bool DM = !(V & 4), RD = !(V & 8), MM = !(V & 16);
    constexpr bool BF = (V & 256) != 0;       // lab ablation (timing only): the B operand costs nothing - no B DMA, no B fragment reads
    const int U = 4 * nk;
    const bool l1 = !TAIL || t + 1 < nk, l2 = !TAIL || t + 2 < nk;
    // ---- phase 0
    nt8_read<RD && !BF>(b0, bP, L);
    nt8_read<RD>(a0, aP, L);
    nt8_read<RD>(a1, aP + 32 * RB, L);
    if (!TAIL) wait_dma_units<3>();
    else wait_dma_units_rt(min(U, 4 * t + 6) - (4 * t + 3));
    nt8_sync_in<V>();
    nt8_mma_issue<T, 2, DM && !BF, MM, SW, (V & 512) != 0>(acc[0][0], acc[1][0], a0, a1, b0, L, smem, t + 1, l1);
    nt8_sync_out<V>();
    // ---- phase 1
    nt8_read<RD && !BF>(b1, bP + 32 * RB, L);
    if (!TAIL) wait_dma_units<3>();
    else wait_dma_units_rt(min(U, 4 * t + 7) - (4 * t + 4));
    nt8_sync_in<V>();
    nt8_mma_issue<T, 3, DM, MM, SW, (V & 512) != 0>(acc[0][1], acc[1][1], a0, a1, b1, L, smem, t + 1, l1);
    nt8_sync_out<V>();
    // ---- phase 2
    nt8_read<RD>(a0, aP + 64 * RB, L);
    nt8_read<RD>(a1, aP + 96 * RB, L);
    nt8_sync_in<V>();
    nt8_mma_issue<T, 0, DM, MM, SW, (V & 512) != 0>(acc[2][1], acc[3][1], a0, a1, b1, L, smem, t + 2, l2);
    nt8_sync_out<V>();
    // ---- phase 3
    if (!TAIL) wait_dma_units<3>();
    else if (t + 1 < nk) wait_dma_units_rt(min(U, 4 * t + 9) - (4 * t + 6));
    nt8_sync_in<V>();
    nt8_mma_issue<T, 1, DM && !BF, MM, SW, (V & 512) != 0>(acc[2][0], acc[3][0], a0, a1, b0, L, smem, t + 2, l2);
    nt8_sync_out<V>();
}

template <typename T, bool TAIL, int V, bool SW>
__device__ __forceinline__ void nt8_ktile(int t, int nk, const NT8Lane& L, char* smem, const char* aP, const char* bP,
                                          f32x16 (&acc)[4][2], i32x4 (&a0)[4], i32x4 (&a1)[4], i32x4 (&b0)[4],
                                          i32x4 (&b1)[4]) {
    constexpr int RB = 128;
    constexpr bool IF = false;
    constexpr bool DM = !(V & 4), RD = !(V & 8), MM = !(V & 16);     // ablations (timing only): no DMA / reads / MFMAs in the loop
    const int U = 4 * nk;
    // ---- phase 0: A sub-tile 0, B fragment 0 -> quadrant (0, 0)
    if (IF && (!TAIL || t + 1 < nk)) nt8_issue<2, DM>(L, smem, t + 1);
    nt8_read<RD>(b0, bP, L);
    nt8_read<RD>(a0, aP, L);
    nt8_read<RD>(a1, aP + 32 * RB, L);
    if (!IF && (!TAIL || t + 1 < nk)) nt8_issue<2, DM>(L, smem, t + 1);
    if (!TAIL) wait_dma_units<4>();
    else wait_dma_units_rt(min(U, 4 * t + 7) - (4 * t + 3));
    nt8_sync_in<V>();
    nt8_mma<T, MM, SW>(acc[0][0], acc[1][0], a0, a1, b0);
    nt8_sync_out<V>();
    // ---- phase 1: B fragment 1 -> quadrant (0, 1)
    if (IF && (!TAIL || t + 1 < nk)) nt8_issue<3, DM>(L, smem, t + 1);
    nt8_read<RD>(b1, bP + 32 * RB, L);
    if (!IF && (!TAIL || t + 1 < nk)) nt8_issue<3, DM>(L, smem, t + 1);
    if (!TAIL) wait_dma_units<4>();
    else wait_dma_units_rt(min(U, 4 * t + 8) - (4 * t + 4));
    nt8_sync_in<V>();
    nt8_mma<T, MM, SW>(acc[0][1], acc[1][1], a0, a1, b1);
    nt8_sync_out<V>();
    // ---- phase 2: A sub-tile 1 -> quadrant (1, 1)
    if (IF && (!TAIL || t + 2 < nk)) nt8_issue<0, DM>(L, smem, t + 2);
    nt8_read<RD>(a0, aP + 64 * RB, L);
    nt8_read<RD>(a1, aP + 96 * RB, L);
    if (!IF && (!TAIL || t + 2 < nk)) nt8_issue<0, DM>(L, smem, t + 2);
    nt8_sync_in<V>();
    nt8_mma<T, MM, SW>(acc[2][1], acc[3][1], a0, a1, b1);
    nt8_sync_out<V>();
    // ---- phase 3: quadrant (1, 0); the wait retires A0 / B0 of K-tile t + 1 for the next phase 0
    if (!TAIL || t + 2 < nk) nt8_issue<1, DM>(L, smem, t + 2);
    if (!TAIL) wait_dma_units<4>();
    else if (t + 1 < nk) wait_dma_units_rt(min(U, 4 * t + 10) - (4 * t + 6));
    nt8_sync_in<V>();
    nt8_mma<T, MM, SW>(acc[2][0], acc[3][0], a0, a1, b0);
    nt8_sync_out<V>();
}


template <typename T, int V, bool SW>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(NTParams p) {
    static_assert(sizeof(T) == 2, "the phased kernel takes the 16-bit storage types");
    constexpr int RB = 128, BM = 256, BN = 256, BK = 64;
    constexpr int kBuf = (BM + BN) * RB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int bm0 = (tile / p.tiles_n) * BM, bn0 = (tile % p.tiles_n) * BN;

    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 0] = wall_clock64();
    NT8Lane L;
    {
        const int lr = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int ra = g * 128 + wid * 8;                           // A0 piece (A1: + 64)
            const int rb = (g * 2 + (wid >> 2)) * 64 + (wid & 3) * 8;   // B0 piece (B1: + 32)
            const int rows[4] = {ra, rb, rb + 32, ra + 64};             // kind 0..3 = A0, B0, B1, A1
#pragma unroll
            for (int kind = 0; kind < 4; ++kind) {
                const int r = rows[kind] + lr;
                const bool isB = (kind == 1 || kind == 2);
                const int64_t grow = isB ? min(bn0 + r, p.N - 1) : min(bm0 + r, p.M - 1);
                L.src[kind][g] = (isB ? p.B + grow * p.ldb : p.A + grow * p.lda) + ((slot ^ lds_swz<RB>(r)) << 4);
                L.dst[kind][g] = (isB ? BM * RB : 0) + rows[kind] * RB;
            }
        }
        const int r = lane & 31, h = lane >> 5, sw = lds_swz<RB>(r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) L.roff[ks] = r * RB + (((ks * 2 + h) ^ sw) << 4);
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // the first chunk's mask words (16 registers) are fetched before anything else: older than every DMA, they retire
    // first and the epilogue finds them in registers instead of waiting an HBM round trip after the last K-tile
    AuxReg<T, 2> pre_bits[16];
    uint32_t row_bits[4][2];
    if constexpr (!SW) {
        if (p.aux_mode == ASE_AUX_RELU_BITS) nt_aux_load<T, 4, 2, 2, 2, 2>(p, 0, lane, bm0 + wr * 128, bn0 + wc * 64, pre_bits);
    }

    const int nk = p.K / BK;
    // prologue: units 0..5 (K-tile 0 and A0, B0 of K-tile 1); A0 / B0 of K-tile 0 must have landed for phase 0
    nt8_issue<0>(L, smem, 0);
    nt8_issue<1>(L, smem, 0);
    nt8_issue<2>(L, smem, 0);
    nt8_issue<3>(L, smem, 0);
    if (nk > 1) {
        nt8_issue<0>(L, smem, 1);
        nt8_issue<1>(L, smem, 1);
    }
    bool mask_dma = false;
    if constexpr (SW) {
        // row-per-lane epilogue: the mask words of the wave tile (128 rows x 2 words) travel as four 4-byte DMA pieces
        // BEHIND the prologue's units into 1 KiB of LDS per wave past the ring (as ordinary loads in front of the DMA
        // queue they add an exposed HBM round trip to the prologue, 3.1 vs 1.4 us; as ordinary loads behind it their
        // position in the vmcnt queue would be the compiler's choice).  Lane (r, h) fetches word h of row 32 i + r;
        // the first counted wait of the main loop retires them.
        mask_dma = p.aux_mode == ASE_AUX_RELU_BITS && bn0 + wc * 64 < p.N;
        if (mask_dma) {
            char* mlds = smem + 2 * kBuf + wid * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = bm0 + wr * 128 + i * 32 + (lane & 31);
                const int ma = (m >= p.aux_split) ? m - p.aux_delta : m;
                const uint32_t* w = reinterpret_cast<const uint32_t*>(p.aux + (int64_t)min(ma, p.M - 1) * p.ldaux) +
                                    ((bn0 + wc * 64) >> 5) + (lane >> 5);
                __builtin_amdgcn_global_load_lds((gptr_t*)w, (lptr_t*)(mlds + i * 256), 4, 0, 0);
            }
        }
    }
    // units 0, 1 (A0 / B0 of K-tile 0) must have landed; the mask pieces (if any) are the 4 youngest entries of the queue
    if (nk > 1) {
        if (mask_dma) wait_vmcnt<8 + 4>(); else wait_dma_units<4>();
    } else {
        if (mask_dma) wait_vmcnt<4 + 4>(); else wait_dma_units<2>();
    }
    NT8_BARRIER();
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 1] = p.prof_clk ? (unsigned long long)clock64() : wall_clock64();
    if (wr == 1) NT8_BARRIER();                  // the second wave group runs one barrier behind

    i32x4 a0[4], a1[4], b0[4], b1[4];
    const int aoff = wr * 128 * RB, boff = BM * RB + wc * 64 * RB;
    int t = 0;
    for (; t + 2 < nk; ++t) {
        const char* buf = smem + (t & 1) * kBuf;
        if constexpr (V & 64) nt8_ktile_m<T, false, V, SW>(t, nk, L, smem, buf + aoff, buf + boff, acc, a0, a1, b0, b1);
        else nt8_ktile<T, false, V, SW>(t, nk, L, smem, buf + aoff, buf + boff, acc, a0, a1, b0, b1);
    }
    for (; t < nk; ++t) {
        const char* buf = smem + (t & 1) * kBuf;
        if constexpr (V & 64) nt8_ktile_m<T, true, V, SW>(t, nk, L, smem, buf + aoff, buf + boff, acc, a0, a1, b0, b1);
        else nt8_ktile<T, true, V, SW>(t, nk, L, smem, buf + aoff, buf + boff, acc, a0, a1, b0, b1);
    }
    if (wr == 0) NT8_BARRIER();
    __syncthreads();                             // the ring becomes the epilogue slab
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 2] = p.prof_clk ? (unsigned long long)clock64() : wall_clock64();

    float* slab = reinterpret_cast<float*>(smem) + wid * (64 * 64);
    if ((V & 32) && p.alpha != 12345.f) return;      // ablation: no epilogue (the guard keeps the accumulators live)
    if constexpr (SW) {
        if (p.aux_mode == ASE_AUX_RELU_BITS) {
            const uint32_t* mw = reinterpret_cast<const uint32_t*>(smem + 2 * kBuf + wid * 1024);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                row_bits[i][0] = mw[i * 64 + (lane & 31)];
                row_bits[i][1] = mw[i * 64 + 32 + (lane & 31)];
            }
            nt8_epilogue_rows<T, 2>(p, acc, lane, bm0 + wr * 128, bn0 + wc * 64, row_bits);
        } else nt8_epilogue_rows<T, 0>(p, acc, lane, bm0 + wr * 128, bn0 + wc * 64, row_bits);
    } else if (p.aux_mode == ASE_AUX_RELU_BITS)
        nt_epilogue_impl<T, 4, 2, 2, 2, 2, true>(p, acc, slab, lane, bm0 + wr * 128, bn0 + wc * 64, &pre_bits);
    else
        nt_epilogue<T, 4, 2, 2, 2>(p, acc, slab, lane, bm0 + wr * 128, bn0 + wc * 64);
    if (p.prof) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) p.prof[blockIdx.x * 4 + 3] = wall_clock64();
    }
}

template <typename T, int V, bool SW = false> int launch_nt8(const NTParams& p0, hipStream_t stream) {
    constexpr int lds = 2 * 512 * 128 + (SW ? 8 * 1024 : 0);     // ring + (row-per-lane epilogue) 1 KiB of mask words per wave
    static bool attr_done = false;
    auto kern = gemm_nt8_kernel<T, V, SW>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            ase_set_error("gemm_nt8: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return ASE_ELAUNCH;
        }
        attr_done = true;
    }
    NTParams p = p0;
    p.prof = g_nt_prof;
    p.prof_clk = g_nt_prof_clk;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    ASE_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), lds, stream, p);
    ASE_CHECK_LAUNCH("gemm_nt8");
    return ASE_OK;
}

// Tuning switches exist only in lab builds (scripts/lab/Makefile compiles this file with -DASE_LAB and reads ASE_* environment
// variables once); the product build has no environment dependence: every knob is its measured default.
#ifdef ASE_LAB
static int lab_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
static constexpr int lab_knob(const char*, int dflt) { return dflt; }
#endif

// Kernel choice of an NT launch (also reported by ase_hip_gemm_nt_kernel_id):
//   0:  64 x  64 tile, 4 waves   narrow heads (N <= 64): more workgroups
//   1: 128 x 128 tile, 4 waves   grids that would leave a 256 x 256 tiling with a ragged round
//   2: 256 x 256 tile, 8 waves, PHASED (16-bit storage, K in whole 128-byte steps)   192+ tiles in whole rounds
//   3: 256 x 256 tile, 8 waves, lock-step (the f32 / bf16x3 storage types)
//   4:  64 x 128 tile, 4 waves / 5: 64 x 64 tile, 4 waves   small grids (M = 2048 ... 4096 rows, or N = 512): two workgroups per CU
int nt_choice(int M, int N, int K, int es, bool b16) {
    static const int force = lab_knob("ASE_NT_TILE", 0), phased = lab_knob("ASE_NT_PHASED", 1);
    if (N <= 64) return 0;
    const int t256 = ((M + 255) / 256) * ((N + 255) / 256);
    const bool big = (force == 256) || (force == 0 && N % 256 == 0 && t256 >= 192 && (t256 <= 256 || t256 % 256 == 0 || t256 >= 1024));
    if (big && force != 128) return (b16 && phased && (K * es) % 128 == 0) ? 2 : 3;
    if (force == 128 || (K * es) % 128 != 0) return 1;
    // grids that do not fill the phased kernel's rounds: the LARGEST of the 128 x 128 / 64 x 128 / 64 x 64 tiles that still
    // gives two workgroups per CU (measured: 4096 x 1024 x 1024  21.1 -> 17.6 us, 4096 x 512 x 1024  19.3 -> 11.4 us,
    // 2048 x 1024 x 1024 - one rank's shard at 8 GPUs -  19.3 -> 11.4 us; 16384 x 512 stays on 128 x 128)
    const int t128 = ((M + 127) / 128) * ((N + 127) / 128), t64x128 = ((M + 63) / 64) * ((N + 127) / 128);
    if (t128 >= 512) return 1;
    if (t64x128 >= 512) return 4;
    return 5;
}

// row-per-lane epilogue (swapped MFMA operands): 16-bit output in whole wave-tile column blocks, no column sums, no tanh,
// mask operand absent or a bit matrix
static bool rows_epi(const NTParams& p, int wave_cols) {
    static const int on = lab_knob("ASE_NT_ROWS", 1);
    return on && !p.out_f32 && p.N % wave_cols == 0 && p.colsum == nullptr && p.act <= ASE_ACT_RELU &&
           (p.aux_mode == ASE_AUX_NONE || p.aux_mode == ASE_AUX_RELU_BITS);
}

template <typename T> int dispatch_nt(const NTParams& p, hipStream_t s) {
    const bool k128 = (p.K * (int)sizeof(T)) % 128 == 0;       // 128-byte staged rows need K in whole 128-byte steps
#ifdef ASE_LAB
    if constexpr (std::is_same<T, bf16_t>::value) {      // (bf16 only: every variant is another kernel instantiation)
        // tuning aid: force one of the co-resident tilings (<= 80 KB of LDS => two workgroups per CU)
        static const int variant = lab_knob("ASE_NT_VARIANT", 0);
        switch (variant) {
            case 10: return launch_nt<T, 2, 2, 2, 4, 64, 3, 2>(p, s);    // 128 x 256, 4 waves (64 x 128 each), 72 KB
            case 11: return launch_nt<T, 2, 2, 4, 2, 64, 3, 2>(p, s);    // 256 x 128, 4 waves (128 x 64 each), 72 KB
            case 12: return launch_nt<T, 4, 2, 2, 2, 64, 3, 4>(p, s);    // 256 x 128, 8 waves (64 x 64 each), 72 KB
            case 13: return launch_nt<T, 2, 4, 2, 2, 64, 3, 4>(p, s);    // 128 x 256, 8 waves
            case 14: if (k128) return launch_nt<T, 2, 2, 2, 2, 128, 2, 2>(p, s); break;   // 128 x 128, 4 waves, 64 KB
            case 15: return launch_nt<T, 2, 2, 2, 2, 64, 4, 2>(p, s);    // 128 x 128, 64-byte rows, 4 stages, 64 KB
            case 16: return launch_nt<T, 2, 2, 2, 2, 64, 3, 3>(p, s);    // 128 x 128, 48 KB => three workgroups per CU
            case 17: return launch_nt<T, 2, 2, 2, 4, 64, 2, 2>(p, s);    // 128 x 256, 4 waves, 2 stages (48 KB => 3 per CU by LDS)
            case 30: if (k128 && rows_epi(p, 128)) return launch_nt<T, 2, 2, 4, 4, 128, 2, 1, true>(p, s); break;   // 256 x 256, FOUR waves
            case 31: if (k128) return launch_nt<T, 2, 2, 4, 4, 128, 2, 1>(p, s); break;
            case 32: if (k128 && rows_epi(p, 64)) return launch_nt<T, 2, 2, 4, 2, 128, 3, 1, true>(p, s); break;   // 256 x 128, four waves, 3-stage ring (144 KB)
            case 20: if (k128) return launch_nt<T, 2, 2, 1, 2, 128, 2, 2>(p, s); break;   // 64 x 128 tile (small M: more workgroups)
            case 21: if (k128) return launch_nt<T, 2, 2, 2, 1, 128, 2, 2>(p, s); break;   // 128 x 64 tile
            case 22: if (k128) return launch_nt<T, 2, 2, 1, 2, 128, 3, 2>(p, s); break;   // 64 x 128, 3 stages
            case 23: if (k128) return launch_nt<T, 2, 2, 1, 1, 128, 4, 2>(p, s); break;   // 64 x 64, 128-byte rows, 4 stages
            // skinny outputs (N <= 64: heads, style columns): HBM-bound streams of A - taller tiles, deeper rings
            case 50: if (k128) return launch_nt<T, 4, 2, 2, 1, 128, 2, 1>(p, s); break;   // 256 x 64, 8 waves, 2 stages (80 KB)
            case 51: if (k128) return launch_nt<T, 4, 1, 2, 2, 128, 3, 1>(p, s); break;   // 256 x 64, 4 waves (64 x 64 each), 3 stages (120 KB)
            case 52: if (k128) return launch_nt<T, 2, 2, 2, 1, 128, 4, 1>(p, s); break;   // 128 x 64, 4 waves, 4 stages (96 KB)
            case 53: if (k128) return launch_nt<T, 2, 2, 2, 1, 128, 3, 2>(p, s); break;   // 128 x 64, 3 stages (72 KB, two per CU)
            default: break;
        }
    }
#endif
    switch (nt_choice(p.M, p.N, p.K, (int)sizeof(T), sizeof(T) == 2)) {
        case 0: return launch_nt<T, 2, 2, 1, 1, 64, 4>(p, s);
        case 2:
            if constexpr (sizeof(T) == 2) {
                // row-per-lane epilogue (swapped MFMA operands): 16-bit output in whole 64-column wave tiles, no column sums,
                // mask operand absent or a bit matrix; otherwise the LDS-slab epilogue.  DMA issued inside the MFMA block (V = 64).
                const bool rows_ok = rows_epi(p, 64);
#ifdef ASE_LAB
                // lab variant: four waves, register-staged operands (scripts/lab/gemm_nt4r_variant.hip)
                static const int nt4r = lab_knob("ASE_NT4R", 0);
                if (nt4r && rows_epi(p, 128) && p.pre_out == nullptr) return launch_nt4r<T>(p, g_nt_prof, s);
#endif
#ifdef ASE_LAB
                if constexpr (std::is_same<T, bf16_t>::value) {      // ablation builds of the phased kernel (timing only)
                    static const int v8 = lab_knob("ASE_NT8_V", -2);
                    if (rows_ok && v8 == 128) return launch_nt8<T, 0, true>(p, s);
                    switch (v8) {
                        case 4: return launch_nt8<T, 4>(p, s);
                        case 8: return launch_nt8<T, 8>(p, s);
                        case 12: return launch_nt8<T, 12>(p, s);
                        case 16: return launch_nt8<T, 16>(p, s);
                        case 28: return launch_nt8<T, 28>(p, s);
                        case 32: return launch_nt8<T, 32>(p, s);
                        case 2: return launch_nt8<T, 2>(p, s);
                        case 1: return launch_nt8<T, 1>(p, s);
                        case 66: return launch_nt8<T, 66>(p, s);
                        case 72: return launch_nt8<T, 72>(p, s);
                        case 80: return launch_nt8<T, 80>(p, s);
                        case 0: return launch_nt8<T, 0>(p, s);
                        case 576: return launch_nt8<T, 576, true>(p, s);     // MFMA-shape ablation: 16 x 16x16x32 per phase (wrong results)
                        case 64: return launch_nt8<T, 64, true>(p, s);       // the product's schedule, for the same-run A/B
                        case 320: return launch_nt8<T, 320, true>(p, s);     // "B operand for free" bound (wrong results)
                        case 328: return launch_nt8<T, 328, true>(p, s);     // ... and no A fragment reads either
                        default: break;
                    }
                }
#endif
                if (rows_ok) return launch_nt8<T, 64, true>(p, s);
                return launch_nt8<T, 64>(p, s);
            }
            [[fallthrough]];
        case 3:
            if (k128) return launch_nt<T, 4, 2, 2, 4, 128, 2>(p, s);
            return launch_nt<T, 4, 2, 2, 4, 64, 4>(p, s);                 // 64-byte rows, 4-stage ring (128 KB)
        case 4:
            if constexpr (sizeof(T) == 2) if (rows_epi(p, 64)) return launch_nt<T, 2, 2, 1, 2, 128, 2, 2, true>(p, s);
            return launch_nt<T, 2, 2, 1, 2, 128, 2, 2>(p, s);            // 64 x 128 tile (K in whole 128-byte steps)
        case 5:
            if constexpr (sizeof(T) == 2) if (rows_epi(p, 32)) return launch_nt<T, 2, 2, 1, 1, 128, 4, 2, true>(p, s);
            return launch_nt<T, 2, 2, 1, 1, 128, 4, 2>(p, s);            // 64 x 64 tile, 128-byte rows, 4 stages
        default:
            if constexpr (sizeof(T) == 2) if (k128 && rows_epi(p, 64)) return launch_nt<T, 2, 2, 2, 2, 128, 2, 1, true>(p, s);
            if (k128) return launch_nt<T, 2, 2, 2, 2, 128, 2>(p, s);
            return launch_nt<T, 2, 2, 2, 2, 64, 4>(p, s);                 // 64-byte rows, 4-stage ring (64 KB)
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

struct TN8Lane {
    const char* src[4][2];     // per-lane DMA source of unit kind (B-lo, A-lo, B-hi, A-hi) x piece, at K-tile 0
    int dst[4][2];             // wave-uniform LDS byte offset of the piece inside a K-tile buffer
    int rbase;                 // per-lane tr-read base: row, 16-byte sub-chunk and half of the lane
    int foffA[4], foffB[2];    // swizzled 64-byte fragment-column offsets
    int64_t kstep[2];          // bytes per K-tile (64 rows) of B / A
};

template <int KIND>
__device__ __forceinline__ void tn8_issue(const TN8Lane& L, char* smem, int tile) {
    char* buf = smem + (tile & 1) * 65536;
    const int64_t koff = (int64_t)tile * L.kstep[KIND & 1];
#pragma unroll
    for (int g = 0; g < 2; ++g)
        __builtin_amdgcn_global_load_lds((gptr_t*)(L.src[KIND][g] + koff), (lptr_t*)(buf + L.dst[KIND][g]), 16, 0, 0);
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
    char* buf = smem + (tile & 1) * 65536;
    const int64_t koff = (int64_t)tile * L.kstep[(KIND < 0 ? 0 : KIND) & 1];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a0[ks] = tn8_join<T>(a[0][ks]);
        a1[ks] = tn8_join<T>(a[1][ks]);
        const x8 b0 = tn8_join<T>(b[0][ks]), b1 = tn8_join<T>(b[1][ks]);
        c00 = mfma16<T>(a0[ks], b0, c00);
        c10 = mfma16<T>(a1[ks], b0, c10);
        if constexpr (KIND >= 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (live)
                __builtin_amdgcn_global_load_lds((gptr_t*)(L.src[KIND][ks] + koff), (lptr_t*)(buf + L.dst[KIND][ks]), 16, 0, 0);
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
template <typename T, bool TAIL, int V, bool BIAS, int HALF>
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
    nt8_sync_in<V>();
    tn8_mma<T, BIAS, HALF == 0 ? 2 : 0>(acc[0][0], acc[0][1], acc[1][0], acc[1][1], bacc, a, b, bias_frag < 2 ? bias_frag : -1,
                                     bvec, L, smem, itile, live);
    nt8_sync_out<V>();
    // ---- odd phase: A fragments 2, 3; the wait retires what the next phase reads (the newest issued unit is the even
    // phase's: three units may stay in flight)
    tn8_read<RO>(a[0][0], adA[2]);
    tn8_read<RO + 8192>(a[0][1], adA[2]);
    tn8_read<RO>(a[1][0], adA[3]);
    tn8_read<RO + 8192>(a[1][1], adA[3]);
    if (!TAIL) wait_dma_units<3>();
    else if (HALF == 0) wait_dma_units_rt(min(U, 4 * t + 7) - (4 * t + 4));
    else if (t + 1 < nk) wait_dma_units_rt(min(U, 4 * t + 9) - (4 * t + 6));
    nt8_sync_in<V>();
    tn8_mma<T, BIAS, HALF == 0 ? 3 : 1>(acc[2][0], acc[2][1], acc[3][0], acc[3][1], bacc, a, b, bias_frag >= 2 ? bias_frag - 2 : -1,
                                     bvec, L, smem, itile, live);
    nt8_sync_out<V>();
}

template <typename T, bool TAIL, int V, bool BIAS>
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
    tn8_half<T, TAIL, V, BIAS, 0>(t, nk, L, smem, adA, adB, bias_frag, bvec, acc, bacc);
    tn8_half<T, TAIL, V, BIAS, 1>(t, nk, L, smem, adA, adB, bias_frag, bvec, acc, bacc);
}

// one work item: output tile (bn0, bk0) of problem p over the K-tiles [m_begin, m_begin + 64 nk)
// ws / wsb (grouped launch): the work item's slab of the partial-sum workspace - the raw accumulator image (64 K floats,
// chunk ((wave * 8 + i * 2 + j) * 4 + g) x 64 lanes x f32x4: every store instruction writes one contiguous KiB) and 256
// bias partial sums - which tn_reduce_kernel folds into the gradient; null: f32 atomics straight into G.
template <typename T, int V>
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
                const char* base = isA ? p.A : p.B;
                const int64_t ld = isA ? p.lda : p.ldb;
                L.src[kind][g] = base + (int64_t)(m_begin + row) * ld + (int64_t)col0 * 2 + chunk * 16;
                L.dst[kind][g] = (isA ? 0 : 32768) + row0 * 512;
            }
        }
        L.kstep[0] = 64 * p.ldb;
        L.kstep[1] = 64 * p.lda;
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
            tn8_ktile<T, false, V, true>(t, nk, L, smem, lds0, wc, t < bias_tiles, acc, bacc);
        for (; t < nk; ++t)
            tn8_ktile<T, true, V, true>(t, nk, L, smem, lds0, wc, t < bias_tiles, acc, bacc);
    } else {
        for (; t + 2 < nk; ++t)
            tn8_ktile<T, false, V, false>(t, nk, L, smem, lds0, wc, false, acc, bacc);
        for (; t < nk; ++t)
            tn8_ktile<T, true, V, false>(t, nk, L, smem, lds0, wc, false, acc, bacc);
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

template <typename T, int V>
__global__ __launch_bounds__(512) void gemm_tn8_kernel(TNParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwg = p.tiles_n * p.tiles_k;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int m_begin = blockIdx.z * p.m_chunk;
    const int m_end = min(p.M, m_begin + p.m_chunk);
    if (m_begin >= m_end) return;
    tn8_body<T, V>(p, smem, (tile / p.tiles_k) * 256, (tile % p.tiles_k) * 256, m_begin, (m_end - m_begin) / 64,
                p.prof ? p.prof + (blockIdx.z * gridDim.x + blockIdx.x) * 4 : nullptr);
}

// Grouped launch.  problems: device int64[n][16] = {A, lda, B, ldb, G, gbias, bias_rows, M, N, K, n_real, k_real,
// split_src, split_dst, alpha (f32 bits), tiles_k}, leading dimensions in ELEMENTS; work: device int32[n_work][4] =
// {problem, tile, m_begin, nk | slab << 16} (ase_hip_gemm_tn_grouped_plan).
template <typename T, int V>
__global__ __launch_bounds__(512) void gemm_tn8g_kernel(const int64_t* __restrict__ problems,
                                                        const int32_t* __restrict__ work, int n_work,
                                                        unsigned long long* prof, float* __restrict__ ws) {
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
    p.alpha = __builtin_bit_cast(float, (int)d[14]);
    p.tiles_k = (int)(d[15] & 0xFFFF);
    tn8_body<T, V>(p, smem, (tile / p.tiles_k) * 256, (tile % p.tiles_k) * 256, m_begin, nk,
                   prof ? prof + blockIdx.x * 4 : nullptr, ws ? ws + (int64_t)slab * kTnSlab : nullptr,
                   ws ? ws + (int64_t)slab * kTnSlab + 65536 : nullptr);
}

// Second kernel of the grouped launch: G += alpha * (sum of the work items' partial tiles), gbias likewise.  One workgroup
// per (reduce entry, quarter tile); red[r] = {problem, tile, first item, splits}, split s of a tile sits `tiles of the
// problem` items further (ase_hip_gemm_tn_grouped_plan's order).  Plain read-modify-write: every (n, k) of a problem has
// exactly one owner; problems that share a gradient buffer with another one (field 15 bit 30 set by the planner: the
// gradient-penalty terms of the encoder land on the discriminator's weights) use atomics.
__global__ __launch_bounds__(256) void tn_reduce_kernel(const int64_t* __restrict__ problems, const int32_t* __restrict__ red,
                                                        const float* __restrict__ ws) {
    const int32_t* r = red + 4 * blockIdx.x;
    const int pi = r[0], tile = r[1], first = r[2], splits = r[3], q = blockIdx.y;
    const int64_t* d = problems + 16 * pi;
    float* G = reinterpret_cast<float*>(d[4]);
    float* gbias = reinterpret_cast<float*>(d[5]);
    const int n_real = (int)d[10], k_real = (int)d[11], split_src = (int)d[12], gap = (int)d[13] - (int)d[12];
    const float alpha = __builtin_bit_cast(float, (int)d[14]);
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

template <typename T, int V> int launch_tn8(TNParams p, hipStream_t stream) {
    constexpr int lds = 2 * 65536;
    static bool attr_done = false;
    auto kern = gemm_tn8_kernel<T, V>;
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

template <typename T, int V> int launch_tn8g(const int64_t* problems, const int32_t* work, int n_work, const int32_t* red,
                                             int n_red, float* ws, hipStream_t stream) {
    constexpr int lds = 2 * 65536;
    static bool attr_done = false;
    auto kern = gemm_tn8g_kernel<T, V>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            ase_set_error("gemm_tn_grouped: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return ASE_ELAUNCH;
        }
        attr_done = true;
    }
    ASE_LAUNCH(kern, dim3(n_work), dim3(512), lds, stream, problems, work, n_work, g_nt_prof, ws);
    if (ws) ASE_LAUNCH(tn_reduce_kernel, dim3(n_red, 16), dim3(256), 0, stream, problems, red, (const float*)ws);
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
    static const int target_env = lab_knob("ASE_TN_TARGET_WG", 0);
    const int target_forced = target_env > 0, target_wg = target_forced ? target_env : 512;
    // narrow outputs (<= 8 tiles): the partial-sum atomics outweigh the second resident workgroup per CU (measured:
    // 256 workgroups beat 512 by 20-30 % on the head / style-MLP gradients)
    int splits = ((tiles <= 8 && !target_forced) ? 256 : target_wg) / tiles;
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

extern "C" int ase_hip_debug_nt_profile(void* buf) {
    g_nt_prof = reinterpret_cast<unsigned long long*>(buf);
    return ASE_OK;
}

extern "C" int ase_hip_debug_nt_profile_clock(int shader_clock) {
    g_nt_prof_clk = shader_clock ? 1 : 0;
    return ASE_OK;
}

extern "C" int ase_hip_gemm_nt_kernel_id(int M, int N, int K, int dtype) {
    return nt_choice(M, N, K, ase_elem_size(dtype), ase_elem_size(dtype) == 2);
}

extern "C" int ase_hip_gemm_nt(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc,
                               const float* bias, const void* aux, int64_t ldaux, int aux_split, int aux_delta,
                               float* colsum, int colsum_n, void* mask_out, int64_t ldmask, int M, int N, int K, int act,
                               int aux_mode, int out_f32, float alpha, int dtype, void* stream) {
    const int es = ase_elem_size(dtype);
    ASE_CHECK_ARG(dtype == ASE_F32 || dtype == ASE_BF16 || dtype == ASE_F32X3 || dtype == ASE_F16, "gemm_nt: bad dtype %d", dtype);
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
    p.M = M; p.N = N; p.K = K; p.act = act; p.aux_mode = aux_mode; p.out_f32 = out_f32; p.alpha = alpha;
    p.tiles_m = p.tiles_n = 0;
    p.prof = nullptr;
    if (dtype == ASE_BF16) return dispatch_nt<bf16_t>(p, (hipStream_t)stream);
    if (dtype == ASE_F16) return dispatch_nt<f16_t>(p, (hipStream_t)stream);
    if (dtype == ASE_F32X3) return dispatch_nt<f32s_t>(p, (hipStream_t)stream);
    return dispatch_nt<float>(p, (hipStream_t)stream);
}

extern "C" int ase_hip_gemm_tn(const void* A, int64_t lda, const void* B, int64_t ldb, float* G, float* gbias,
                               int bias_rows, int M, int N, int K, int n_real, int k_real, int split_src, int split_dst, float alpha, int dtype,
                               void* stream) {
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
    p.alpha = alpha; p.tiles_n = p.tiles_k = p.m_chunk = 0; p.prof = nullptr;
    if (es == 2) {
        // Single-problem launches take the phased 256 x 256 kernel only when few M-splits fill the chip (its split
        // reduction costs 256 KB of memory-side atomics per workgroup; see the grouped launch): whole 64-row K-tiles,
        // whole bias tiles, >= 32 K-tiles per split.
        static const int tn8 = lab_knob("ASE_TN8", 1);
        const int t256 = ((n_real + 255) / 256) * ((K + 255) / 256);
        const bool phased = tn8 && M % 64 == 0 && p.bias_rows % 64 == 0 && n_real >= 128 && K >= 128 && (int64_t)M * t256 >= 256 * 2048;
        if (dtype == ASE_BF16) return phased ? launch_tn8<bf16_t, 0>(p, (hipStream_t)stream) : launch_tn<bf16_t>(p, (hipStream_t)stream);
        return phased ? launch_tn8<f16_t, 0>(p, (hipStream_t)stream) : launch_tn<f16_t>(p, (hipStream_t)stream);
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
    return ASE_OK;
}

extern "C" int ase_hip_gemm_tn_grouped_plan(int64_t* problems, int n_problems, int target_wg, int32_t* work, int max_work,
                                            int* n_work, int32_t* red, int max_red, int* n_red) {
    ASE_CHECK_ARG(problems && work && n_work && n_problems > 0 && max_work > 0, "gemm_tn_grouped_plan: null/empty argument");
    ASE_CHECK_ARG(red == nullptr || (n_red && max_red > 0), "gemm_tn_grouped_plan: reduce table without its size");
    if (target_wg <= 0) target_wg = lab_knob("ASE_TN_GROUP_WG", 256);      // one 8-wave workgroup per CU
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
                                       float* workspace, int dtype, void* stream) {
    ASE_CHECK_ARG(problems && work && n_work > 0, "gemm_tn_grouped: null/empty argument");
    ASE_CHECK_ARG(dtype == ASE_BF16 || dtype == ASE_F16, "gemm_tn_grouped: 16-bit storage types only (dtype %d)", dtype);
    ASE_CHECK_ARG(workspace == nullptr || (red && n_red > 0 && ((uintptr_t)workspace % 16) == 0),
                  "gemm_tn_grouped: a workspace needs the reduce table of the plan (and 16-byte alignment)");
    if (dtype == ASE_F16) return launch_tn8g<f16_t, 0>(problems, work, n_work, red, n_red, workspace, (hipStream_t)stream);
    return launch_tn8g<bf16_t, 0>(problems, work, n_work, red, n_red, workspace, (hipStream_t)stream);
}

extern "C" int ase_hip_refresh_shadow(const float* W, int n_real, int k_real, void* Ws, int64_t ldws, void* Wts,
                                      int64_t ldwts, int split_src, int split_dst, int dtype, void* stream) {
    ASE_CHECK_ARG(W && n_real > 0 && k_real > 0 && (Ws || Wts), "refresh_shadow: null/empty operand");
    ASE_CHECK_ARG(split_src <= split_dst && split_src <= k_real, "refresh_shadow: bad split");
    const dim3 grid((k_real + 31) / 32, (n_real + 31) / 32), block(256);
    const int gap = split_dst - split_src;
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
