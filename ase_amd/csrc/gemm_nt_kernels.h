// NT matrix-core kernels of the dense layers (gfx950 / CDNA4): C = mask(act(alpha * A·Bᵀ + bias)) - forward, data-gradient,
// gradient-penalty chain.  Templates only; each storage type is instantiated in its own translation unit
// (gemm_nt_<type>.hip -> ase_nt::dispatch_nt_<type>) so that the library builds in parallel.
//
// Storage types: bf16 / f16 (v_mfma_f32_32x32x16_{bf16,f16}, f32 accumulate), exact f32 (v_mfma_f32_32x32x2_f32), f32 multiplied
// as three bf16 MFMAs on a hi/lo split (f32s_t) and f32 multiplied as three f16 MFMAs on a hi/lo split of SCALED operands
// (f32h_t).  Wave64, XCD-aware tile order (8 XCDs, private L2s).
//   Staging: tiles go HBM -> LDS directly with global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass); LDS rows
//       are 128 bytes, unpadded (the DMA writes lane-linear), with the 16-byte chunks of row r stored at slot
//       chunk ^ ((r >> 1) & 7): the permutation is applied to the per-lane SOURCE address and again on the fragment
//       read, which makes every ds_read_b128 lane group hit 16 distinct bank slots.
//   Kernels: gemm_nt_kernel (64 / 128 / 256 tiles, S-stage ring, one barrier per K-tile, every wave in lock-step) and
//       gemm_nt8_kernel (16-bit storage, 256 x 256: four phases per K-tile, two wave groups one barrier apart, counted
//       vmcnt) - see the comment blocks in front of each; nt_choice() picks per shape.
#pragma once
#include "gemm_nt.h"
#include <stdlib.h>

namespace {

using namespace ase_nt;

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

// f32h_t (ASE_F32H3): f32-sized storage whose products run as three f16 MFMAs on hi / lo splits of power-of-two SCALED operands.
//   A (activations, chain values: f32 in HBM and LDS) is split in registers:  sx = x * sa (exact), hi = half(sx), lo = half(sx - hi).
//     With |sx| >= 2^-2 both parts are normal halves and hi + lo carries 22 significant bits of x; smaller values degrade gracefully
//     (lo subnormal: absolute error 2^-25 of the scaled range); |sx| > 65504 overflows to inf and the launch's output is NaN - loud
//     on purpose (the caller's scale exponent was wrong), never a silently saturated value.
//   B (weights) arrives PRE-SPLIT: ase_hip_refresh_shadow(dtype = ASE_F32H3 | eb << 16) writes the shadows in the packed format
//     [8 hi halves | 8 lo halves] per group of 8 consecutive k (32 bytes, the footprint of 8 floats), already scaled by 2^eb.  A B
//     element is needed by every row tile of the launch (64 of them at 4096 rows): splitting it once per optimisation step instead
//     of once per tile removes two thirds of the kernel's VALU work, which is what bounded it (round 5: 5 VALU operations per
//     element and fragment against 3 MFMAs of 32 cycles per fragment pair).
__device__ __forceinline__ void split_f16(const f32x4& x0, const f32x4& x1, float scale, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float s0 = x0[q] * scale, s1 = x1[q] * scale;
        hi[q] = (f16_t)s0;
        hi[q + 4] = (f16_t)s1;
        lo[q] = (f16_t)(s0 - (float)hi[q]);
        lo[q + 4] = (f16_t)(s1 - (float)hi[q + 4]);
    }
}

template <> struct Mma<f32h_t> {
    // rows of 128 / 64 bytes = 32 / 16 k-values; per k-step of 16 a lane (r, h) owns the 8 k-values of group 2 ks + h = two
    // 16-byte chunks: A - 8 floats to split; B - its 8 hi halves and its 8 lo halves
    template <int FM, int FN, int RB>
    static __device__ __forceinline__ void tile(const char* sA, const char* sB, int lane, f32x16 (&acc)[FM][FN], float sa) {
        const int r = lane & 31, h = lane >> 5, sw = lds_swz<RB>(r);
#pragma unroll
        for (int ks = 0; ks < RB / 64; ++ks) {
            f16x8 ah[FM], al[FM], bh[FN], bl[FN];
            const int c0 = ks * 4 + h * 2;
            const int o0 = r * RB + ((c0 ^ sw) << 4), o1 = r * RB + (((c0 + 1) ^ sw) << 4);
#pragma unroll
            for (int i = 0; i < FM; ++i)
                split_f16(*reinterpret_cast<const f32x4*>(sA + i * 32 * RB + o0),
                          *reinterpret_cast<const f32x4*>(sA + i * 32 * RB + o1), sa, ah[i], al[i]);
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(sB + j * 32 * RB + o0);
                bl[j] = *reinterpret_cast<const f16x8*>(sB + j * 32 * RB + o1);
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
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
template <> struct AuxReg<f32h_t, 1> { f32x4 v; };
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
    bool bad = false;                              // an element this lane stored overflowed (scale records, common.h)

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
                            for (int q = 0; q < 4; ++q) {
                                zt[q] = from_f32<T>(v[q]);
                                bad |= ovf_hit1(zt[q]);
                            }
                            *reinterpret_cast<typename V16<T>::x4*>(p.pre_out + (int64_t)m * p.ldpre + (int64_t)n0 * 2) = zt;
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) bad |= ovf_hit1(v[q]);
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
#pragma unroll
                    for (int q = 0; q < 4; ++q) bad |= ovf_hit1(v[q]);
                    *reinterpret_cast<f32x4*>(p.C + (int64_t)m * p.ldc + (int64_t)n0 * 4) = v;
                } else {
                    typedef typename std::conditional<sizeof(T) == 2, T, bf16_t>::type S;     // (4-byte T: dead branch)
                    typename V16<S>::x4 o;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[q] = from_f32<S>(v[q]);
                        bad |= ovf_hit1(o[q]);
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
    ovf_report(p.alpha_dev, bad);
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
template <typename T, int FM, int FN, int AUXK, bool MASK>
__device__ __forceinline__ void nt_epilogue_rows_impl(const NTParams& p, f32x16 (&acc)[FM][FN], int lane, int mrow0, int ncol0,
                                                      const uint32_t (&bits)[FM][FN]) {
    const int r = lane & 31, h = lane >> 5;
    const f32x2 al = {p.alpha, p.alpha};
    const short lo = (p.act == ASE_ACT_RELU) ? (short)0 : (short)-32768;
    const i16x2 relu_lo = {lo, lo};
    uint32_t ovf = 0;
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
            const uint32_t mb = rows_frag<T, AUXK, MASK>(acc[i][j], bias, AUXK == 2 ? bits[i][j] >> (4 * h) : 0u, al, relu_lo, h, pk);
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                if (row_ok) *reinterpret_cast<uint4*>(crow + 8 * g * 2) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                ovf = ovf_fold(ovf_fold(ovf_fold(ovf_fold(ovf, s0[0]), s1[0]), s0[1]), s1[1]);
            }
            if constexpr (MASK) {
                const auto w = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);   // own 16 bits | the other half-wave's
                if (row_ok && h == 0) p.mask_out[(int64_t)m * p.ldmask + ((ncol0 + j * 32) >> 5)] = w[0] | w[1];
            }
        }
    }
    ovf_report(p.alpha_dev, ovf_hit<T>(ovf));
}

template <typename T, int FM, int FN, int AUXK>
__device__ __forceinline__ void nt_epilogue_rows(const NTParams& p, f32x16 (&acc)[FM][FN], int lane, int mrow0, int ncol0,
                                                 const uint32_t (&bits)[FM][FN]) {
    if (ncol0 >= p.N) return;                                   // wave-uniform: N is a multiple of the wave tile's width
    if (p.mask_out) nt_epilogue_rows_impl<T, FM, FN, AUXK, true>(p, acc, lane, mrow0, ncol0, bits);
    else nt_epilogue_rows_impl<T, FM, FN, AUXK, false>(p, acc, lane, mrow0, ncol0, bits);
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
    if (p.alpha_dev) p.alpha *= *p.alpha_dev;

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
        else if constexpr (std::is_same<T, f32h_t>::value) Mma<T>::template tile<FM, FN, RB>(sA, sB, lane, acc, p.sa);
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
// NT, phased 256 x 256 kernel (16-bit storage).  512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 outputs =
// four 64 x 32 quadrants.  One K-tile (64 k-values, 128-byte rows, the swizzle of the kernel above) is FOUR phases, each
//     ds_read the fragments of one quadrant | counted vmcnt | s_barrier | lgkmcnt(0) | 8 MFMAs 32x32x16 with the two
//     global_load_lds of one 16-KiB DMA unit issued among them | s_barrier
// and the two wave groups (waves 0-3 / 4-7: the two waves of every SIMD sit in different groups) run ONE BARRIER
// APART, so that on each SIMD one wave is in its MFMA block while its partner reads LDS.
//   DMA units of K-tile t, in issue order = order of first use:
//     A0 = A rows {0-63, 128-191} (sub-tile 0 of both wave rows)     read in phase 0
//     B0 = B rows {64 c .. 64 c + 31, c = 0..3} (fragment 0 of every wave column)   phase 0 (kept in registers to phase 3)
//     B1 = B rows {64 c + 32 .. 64 c + 63}                            phase 1
//     A1 = A rows {64-127, 192-255}                                   phase 2
//   unit u = 4 t + kind is issued in phase (t', p) with 4 t' + p + 6 = u: six units ahead, into the buffer (t & 1) whose
//   previous occupant (K-tile t - 2) was last read >= 2 phases earlier (the WAR distance two staggered groups need);
//   a unit is read one phase after the counted wait + barrier that retires it (RAW across the stagger).  The DMA of a phase
//   goes out INSIDE its MFMA block (an LDS-DMA instruction costs ~60 issue cycles beside MFMAs - the matrix pipe stays fed
//   by the 32-cycle MFMA issue cadence - but 100-185 cycles in the read half of a phase, where it sat on the critical path
//   of the OTHER wave group's MFMA block): at the counted wait of a phase the newest issued unit is the one of the previous
//   phase, three units may stay in flight.
//   LDS balance (round 5, measured and NOT adopted): the fragment reads are 12 / 4 / 8 / 0 KB per wave over the four phases; reading
//   B fragment 0 of the next K-tile in phase 3 instead (8 / 4 / 8 / 4, 12 more registers) left the main loop where it was - 41831
//   against 41708 shader cycles for 16 K-tiles of 16384 x 1024 x 1024 (profiles/r05_nt8_lds_rebalance.txt): a phase costs ~650
//   cycles for 2 x 256 of MFMA issue, and the ~70 cycles per hand-over between the two wave groups (barrier release, lgkmcnt,
//   priority switch, the DMA issue inside the block) are what is left, not the LDS port.
//   (Round 2-4 ablations of this schedule - no DMA / no reads / no MFMAs / other MFMA shapes / DMA in the read half - lived
//   behind a lab switch in this file up to commit 9f99095; their results are in profiles/r03_lab_*.log, r04_lab_mfma_shape.txt.)
// ------------------------------------------------------------------------------------------------

// DMA pieces go out as `buffer_load_dwordx4 ... lds` (round 6, like gemm_nt4.h): an SGPR buffer descriptor per operand, ONE 32-bit
// offset register per piece (the 64-bit per-lane addresses of global_load_lds cost 16 registers and two VALU adds per piece), the
// K-tile's byte offset as the instruction's SGPR offset, M0 = LDS destination.  What it buys is REGISTERS: the kernel's allocation
// decides how many of the 512 registers per SIMD its two waves leave to the small HBM-bound kernels of the step's other branches,
// which run in exactly that space (profiles/r06_nt_policy_ab.txt: a matrix kernel that owns the whole file costs the update 2.2 %).
struct NT8Lane {
    uint32_t voff[4][2];       // per-lane byte offset (row * ld + swizzled chunk) of unit kind x piece from its operand's base
    uint32_t dst[4][2];        // wave-uniform LDS byte ADDRESS of the piece in K-tile buffer 0 (buffer 1: + 65536)
    int roff[4];               // per-lane fragment read offsets (row * 128 + swizzled chunk) for the 4 k-steps
    i32x4 rsA, rsB;            // raw buffer descriptors of the two operands (SGPRs)
};

__device__ __forceinline__ void nt8_dma(uint32_t lds_s, uint32_t voff, i32x4 rs, uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_s), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
template <int KIND>
__device__ __forceinline__ void nt8_piece(const NT8Lane& L, int g, int tile) {
    constexpr bool isB = (KIND == 1 || KIND == 2);
    nt8_dma(__builtin_amdgcn_readfirstlane(L.dst[KIND][g] + (uint32_t)(tile & 1) * 65536u), L.voff[KIND][g], isB ? L.rsB : L.rsA,
            (uint32_t)tile * 128u);
}

template <int KIND>
__device__ __forceinline__ void nt8_issue(const NT8Lane& L, char* smem, int tile) {
#pragma unroll
    for (int g = 0; g < 2; ++g) nt8_piece<KIND>(L, g, tile);
}

// fragment registers of one 32-row operand block: 4 k-steps x 16 bytes
__device__ __forceinline__ void nt8_read(i32x4 (&f)[4], const char* base, const NT8Lane& L) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f[ks] = *reinterpret_cast<const i32x4*>(base + L.roff[ks]);
}

// the 8 MFMAs of a phase with the two DMA pieces of unit KIND (K-tile `tile`) issued among them, and the phase's CLOSING barrier
// in front of the last MFMA pair: every s_barrier of the loop is the end of one wave group's MFMA block and the start of the
// other's, and between the first group's last MFMA issue and the second group's first one lie the barrier's release, an lgkmcnt
// wait and a priority switch (~70 of the ~650 cycles a phase took, round 5's cycle stamps).  With two MFMAs (64 cycles of
// matrix-pipe work) still to issue behind the barrier, the pipe stays fed across the hand-over.  Nothing those two MFMAs touch
// is shared: their operands are in registers since the phase's first barrier.
template <typename T, int KIND, bool SW>
__device__ __forceinline__ void nt8_mma_issue(f32x16& c0, f32x16& c1, const i32x4 (&a0)[4], const i32x4 (&a1)[4],
                                              const i32x4 (&b)[4], const NT8Lane& L, char* smem, int tile, bool live) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ks == 3) NT8_BARRIER();                       // (closes the phase: see above)
        c0 = nt8_mfma<T, SW>(a0[ks], b[ks], c0);
        c1 = nt8_mfma<T, SW>(a1[ks], b[ks], c1);
        if (ks == 0 || ks == 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (live) nt8_piece<KIND>(L, ks >> 1, tile);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
}

// ... and its half-height twin for the 192 x 256 tile's third A block: 4 MFMAs on one accumulator, same DMA / barrier placement
template <typename T, int KIND, bool SW>
__device__ __forceinline__ void nt8_mma_issue1(f32x16& c0, const i32x4 (&a0)[4], const i32x4 (&b)[4], const NT8Lane& L, char* smem,
                                               int tile, bool live) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        if (ks == 3) NT8_BARRIER();
        c0 = nt8_mfma<T, SW>(a0[ks], b[ks], c0);
        if (ks == 0 || ks == 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (live) nt8_piece<KIND>(L, ks >> 1, tile);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
}

// one K-tile = 4 phases.  TAIL = false: every issued unit exists (t + 2 < nk) and the waits are compile-time counts.
// NI = 4: wave tile 128 x 64 (256 x 256 tile); NI = 3: wave tile 96 x 64 (192 x 256 tile) - phases 2 and 3 multiply ONE A block
template <typename T, bool TAIL, bool SW, int NI = 4>
__device__ __forceinline__ void nt8_ktile(int t, int nk, const NT8Lane& L, char* smem, const char* aP, const char* bP,
                                          f32x16 (&acc)[NI][2], i32x4 (&a0)[4], i32x4 (&a1)[4], i32x4 (&b0)[4],
                                          i32x4 (&b1)[4]) {
    constexpr int RB = 128;
    const int U = 4 * nk;
    const bool l1 = !TAIL || t + 1 < nk, l2 = !TAIL || t + 2 < nk;
    // ---- phase 0: A sub-tile 0, B fragment 0 -> quadrant (0, 0); issues B1 of K-tile t + 1
    nt8_read(b0, bP, L);
    nt8_read(a0, aP, L);
    nt8_read(a1, aP + 32 * RB, L);
    if (!TAIL) wait_dma_units<3>();
    else wait_dma_units_rt(min(U, 4 * t + 6) - (4 * t + 3));
    nt8_sync_in();
    nt8_mma_issue<T, 2, SW>(acc[0][0], acc[1][0], a0, a1, b0, L, smem, t + 1, l1);
    // ---- phase 1: B fragment 1 -> quadrant (0, 1); issues A1 of K-tile t + 1
    nt8_read(b1, bP + 32 * RB, L);
    if (!TAIL) wait_dma_units<3>();
    else wait_dma_units_rt(min(U, 4 * t + 7) - (4 * t + 4));
    nt8_sync_in();
    nt8_mma_issue<T, 3, SW>(acc[0][1], acc[1][1], a0, a1, b1, L, smem, t + 1, l1);
    // ---- phase 2: A sub-tile 1 -> quadrant (1, 1); issues A0 of K-tile t + 2
    nt8_read(a0, aP + 64 * RB, L);
    if constexpr (NI == 4) nt8_read(a1, aP + 96 * RB, L);
    nt8_sync_in();
    if constexpr (NI == 4) nt8_mma_issue<T, 0, SW>(acc[2][1], acc[3][1], a0, a1, b1, L, smem, t + 2, l2);
    else nt8_mma_issue1<T, 0, SW>(acc[2][1], a0, b1, L, smem, t + 2, l2);
    // ---- phase 3: quadrant (1, 0); issues B0 of K-tile t + 2; the wait retires A0 / B0 of K-tile t + 1 for the next phase 0
    if (!TAIL) wait_dma_units<3>();
    else if (t + 1 < nk) wait_dma_units_rt(min(U, 4 * t + 9) - (4 * t + 6));
    nt8_sync_in();
    if constexpr (NI == 4) nt8_mma_issue<T, 1, SW>(acc[2][0], acc[3][0], a0, a1, b0, L, smem, t + 2, l2);
    else nt8_mma_issue1<T, 1, SW>(acc[2][0], a0, b0, L, smem, t + 2, l2);
}


// SW: swapped MFMA operands + row-per-lane epilogue (nt8_epilogue_rows), else the LDS-slab epilogue
// BM = 192 (row-per-lane epilogue only): the same schedule on a 192 x 256 tile - wave tile 96 x 64, phases 2 / 3 are 4 MFMAs - for
// row counts whose 256-row tiling leaves a quarter of the CUs idle (12288 x 1024: 192 tiles of 256 rows, 256 tiles of 192).  The
// K-tile buffers keep their 64-KiB stride and B keeps following A (offset BM * 128); unit A1 covers rows 64-95 of BOTH wave rows
// with one piece, its second piece repeats the first (same bytes, same place) so that the counted waits stay what they are.
template <typename T, bool SW, int BM = 256>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(NTParams p) {
    static_assert(sizeof(T) == 2, "the phased kernel takes the 16-bit storage types");
    static_assert(BM == 256 || (BM == 192 && SW), "192-row tiles come with the row-per-lane epilogue");
    constexpr int RB = 128, BN = 256, BK = 64, WM = BM / 2, NI = WM / 32;
    constexpr int kBuf = 512 * RB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.alpha_dev) p.alpha *= *p.alpha_dev;

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
            const int ra = g * WM + wid * 8;                            // A0 piece (A1 of the 256-row tile: + 64)
            const int rb = (g * 2 + (wid >> 2)) * 64 + (wid & 3) * 8;   // B0 piece (B1: + 32)
            const int ra1 = BM == 256 ? ra + 64 : (wid >> 2) * WM + 64 + (wid & 3) * 8;
            const int rows[4] = {ra, rb, rb + 32, ra1};                 // kind 0..3 = A0, B0, B1, A1
#pragma unroll
            for (int kind = 0; kind < 4; ++kind) {
                const int r = rows[kind] + lr;
                const bool isB = (kind == 1 || kind == 2);
                const int64_t grow = isB ? min(bn0 + r, p.N - 1) : min(bm0 + r, p.M - 1);
                L.voff[kind][g] = (uint32_t)(grow * (isB ? p.ldb : p.lda)) + (uint32_t)((slot ^ lds_swz<RB>(r)) << 4);
                L.dst[kind][g] = (uint32_t)(uintptr_t)smem + (isB ? BM * RB : 0) + rows[kind] * RB;
            }
        }
        const int r = lane & 31, h = lane >> 5, sw = lds_swz<RB>(r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) L.roff[ks] = r * RB + (((ks * 2 + h) ^ sw) << 4);
        // whole operands as raw buffers (rows past M / N are clamped in the per-lane offsets above)
        const uint64_t a = (uint64_t)(uintptr_t)p.A, b = (uint64_t)(uintptr_t)p.B;
        L.rsA = i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)0x7FFFFFFF, 0x00020000};
        L.rsB = i32x4{(int)(uint32_t)b, (int)(uint32_t)(b >> 32), (int)0x7FFFFFFF, 0x00020000};
    }

    f32x16 acc[NI][2];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // the first chunk's mask words (16 registers) are fetched before anything else: older than every DMA, they retire
    // first and the epilogue finds them in registers instead of waiting an HBM round trip after the last K-tile
    AuxReg<T, 2> pre_bits[16];
    uint32_t row_bits[NI][2];
    if constexpr (!SW) {
        if (p.aux_mode == ASE_AUX_RELU_BITS) nt_aux_load<T, 4, 2, 2, 2, 2>(p, 0, lane, bm0 + wr * WM, bn0 + wc * 64, pre_bits);
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
            for (int i = 0; i < NI; ++i) {
                const int m = bm0 + wr * WM + i * 32 + (lane & 31);
                const int ma = (m >= p.aux_split) ? m - p.aux_delta : m;
                const uint32_t* w = reinterpret_cast<const uint32_t*>(p.aux + (int64_t)min(ma, p.M - 1) * p.ldaux) +
                                    ((bn0 + wc * 64) >> 5) + (lane >> 5);
                __builtin_amdgcn_global_load_lds((gptr_t*)w, (lptr_t*)(mlds + i * 256), 4, 0, 0);
            }
        }
    }
    // units 0, 1 (A0 / B0 of K-tile 0) must have landed; the mask pieces (if any) are the NI youngest entries of the queue
    if (nk > 1) {
        if (mask_dma) wait_vmcnt<8 + NI>(); else wait_dma_units<4>();
    } else {
        if (mask_dma) wait_vmcnt<4 + NI>(); else wait_dma_units<2>();
    }
    NT8_BARRIER();
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 1] = p.prof_clk ? (unsigned long long)clock64() : wall_clock64();
    if (wr == 1) NT8_BARRIER();                  // the second wave group runs one barrier behind

    i32x4 a0[4], a1[4], b0[4], b1[4];
    const int aoff = wr * WM * RB, boff = BM * RB + wc * 64 * RB;
    int t = 0;
    for (; t + 2 < nk; ++t) {
        const char* buf = smem + (t & 1) * kBuf;
        nt8_ktile<T, false, SW, NI>(t, nk, L, smem, buf + aoff, buf + boff, acc, a0, a1, b0, b1);
    }
    for (; t < nk; ++t) {
        const char* buf = smem + (t & 1) * kBuf;
        nt8_ktile<T, true, SW, NI>(t, nk, L, smem, buf + aoff, buf + boff, acc, a0, a1, b0, b1);
    }
    if (wr == 0) NT8_BARRIER();
    __syncthreads();                             // the ring becomes the epilogue slab
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 2] = p.prof_clk ? (unsigned long long)clock64() : wall_clock64();

    float* slab = reinterpret_cast<float*>(smem) + wid * (64 * 64);
    if constexpr (SW) {
        if (p.aux_mode == ASE_AUX_RELU_BITS) {
            const uint32_t* mw = reinterpret_cast<const uint32_t*>(smem + 2 * kBuf + wid * 1024);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                row_bits[i][0] = mw[i * 64 + (lane & 31)];
                row_bits[i][1] = mw[i * 64 + 32 + (lane & 31)];
            }
            nt8_epilogue_rows<T, 2, 2, NI>(p, acc, lane, bm0 + wr * WM, bn0 + wc * 64, row_bits);
        } else nt8_epilogue_rows<T, 0, 2, NI>(p, acc, lane, bm0 + wr * WM, bn0 + wc * 64, row_bits);
    } else if constexpr (BM == 256) {
        if (p.aux_mode == ASE_AUX_RELU_BITS)
            nt_epilogue_impl<T, 4, 2, 2, 2, 2, true>(p, acc, slab, lane, bm0 + wr * 128, bn0 + wc * 64, &pre_bits);
        else
            nt_epilogue<T, 4, 2, 2, 2>(p, acc, slab, lane, bm0 + wr * 128, bn0 + wc * 64);
    }
    if (p.prof) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) p.prof[blockIdx.x * 4 + 3] = wall_clock64();
    }
}

template <typename T, bool SW = false, int BM = 256> int launch_nt8(const NTParams& p0, hipStream_t stream) {
    constexpr int lds = 2 * 512 * 128 + (SW ? 8 * 1024 : 0);     // ring + (row-per-lane epilogue) 1 KiB of mask words per wave
    static bool attr_done = false;
    auto kern = gemm_nt8_kernel<T, SW, BM>;
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
    p.tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + 255) / 256;
    ASE_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n), dim3(512), lds, stream, p);
    ASE_CHECK_LAUNCH("gemm_nt8");
    return ASE_OK;
}

}  // namespace
#include "gemm_nt4.h"
namespace {

// row-per-lane epilogue (swapped MFMA operands): 16-bit output in whole wave-tile column blocks, no column sums, no tanh,
// mask operand absent or a bit matrix
inline bool rows_epi(const NTParams& p, int wave_cols) {
    return !p.out_f32 && p.N % wave_cols == 0 && p.colsum == nullptr && p.act <= ASE_ACT_RELU &&
           (p.aux_mode == ASE_AUX_NONE || p.aux_mode == ASE_AUX_RELU_BITS);
}

template <typename T> int dispatch_nt(const NTParams& p, hipStream_t s) {
    const bool k128 = (p.K * (int)sizeof(T)) % 128 == 0;       // 128-byte staged rows need K in whole 128-byte steps
    // (the phased kernels and the 4-wave kernel address their operands with 32-bit byte offsets from the base: < 2 GiB each)
    const bool fits32 = (int64_t)p.M * p.lda < (int64_t)0x7FFFFFFF && (int64_t)p.N * p.ldb < (int64_t)0x7FFFFFFF;
    int choice = nt_choice(p.M, p.N, p.K, (int)sizeof(T), sizeof(T) == 2);
    if ((choice == 2 || choice == 6) && !fits32) choice = 3;
    switch (choice) {
        case 0: return launch_nt<T, 2, 2, 1, 1, 64, 4>(p, s);
        case 6:
            if constexpr (sizeof(T) == 2) if (rows_epi(p, 64)) return launch_nt8<T, true, 192>(p, s);
            [[fallthrough]];
        case 2:
            if constexpr (sizeof(T) == 2) {
                // row-per-lane epilogue (swapped MFMA operands): 16-bit output in whole 64-column wave tiles, no column sums,
                // mask operand absent or a bit matrix; otherwise the LDS-slab epilogue
                // The 4-wave kernel (gemm_nt4.h: whole 128-column wave tiles, operands a 32-bit byte offset reaches) takes the launches of
                // MANY rounds that run with nothing beside them - the epoch tail's reward inference over the whole experience buffer
                // (131072 rows: 324 vs 344 us).  Inside an optimisation step it stays out although it is 4-8 % faster launch by launch
                // (profiles/r06_nt4_lab.txt): its workgroup owns the CU's whole register file (4 waves x 512), the 8-wave kernel leaves
                // 32 registers per SIMD, and in the four-stream step the small HBM-bound kernels of the other branches (<= 32 VGPRs:
                // gathers, conversions, loss heads) run in exactly that space - same-box A/B of the update: 77.5 ms with the 8-wave
                // kernel, 79.3 ms with this one (profiles/r06_nt_policy_ab.txt).
                if (p.M >= 65536 && rows_epi(p, 128)) return launch_nt4<T>(p, s);
                if (rows_epi(p, 64)) return launch_nt8<T, true>(p, s);
                return launch_nt8<T, false>(p, s);
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

}  // namespace
