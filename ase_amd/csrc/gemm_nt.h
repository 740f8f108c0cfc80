// Pieces of the NT matrix-core kernels (gfx950) kept in a header: tile order, LDS swizzle, launch
// parameters, the MFMA wrapper with swapped operands and the row-per-lane epilogue of the phased kernels.
#pragma once
#include "common.h"
#include "act.h"
#include <type_traits>

namespace ase_nt {


// Bijective XCD-aware remap: workgroup b runs on XCD b % 8; give each XCD a contiguous tile range.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}


// LDS rows are RB = 64 or 128 bytes, unpadded; chunk c (16 B) of row r lives at slot c ^ swz(r):
//   RB = 128: swz = (r >> 1) & 7   (a 256-byte bank window holds 2 rows x 8 slots)
//   RB =  64: swz = (r >> 2) & 3   (4 rows x 4 slots)
// either way the 16 rows (distinct mod 16) of a ds_read_b128 lane group land on 16 distinct slots.
template <int RB> __device__ __forceinline__ int lds_swz(int r) { return RB == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3); }

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct NTParams {
    const char* A; int64_t lda;     // leading dims in BYTES
    const char* B; int64_t ldb;
    char* C; int64_t ldc;           // bytes
    const float* bias;
    const char* aux; int64_t ldaux; // bytes
    int aux_split, aux_delta;       // rows m >= aux_split read aux row m - aux_delta (stacked row blocks sharing a mask)
    float* colsum; int colsum_n;
    uint32_t* mask_out; int64_t ldmask;   // nullable: bit (m, n) = stored value > 0, 32 columns per word, ldmask in words
    char* pre_out; int64_t ldpre;         // nullable (smooth activations): the pre-activation z in the storage type, ldpre in bytes
    int M, N, K;                    // K in elements (multiple of 128/sizeof(T))
    int act, aux_mode, out_f32;
    float alpha;
    float* alpha_dev;               // nullable: device record {factor, overflow count} (ase_hip.h "scale records"): factor is multiplied into
                                    // alpha when the launch runs (the dynamic loss scale), and the launch adds to the count when an
                                    // element it stored was non-finite or sat at the storage type's saturation value
    float sa;                       // f32h_t storage: power-of-two scale of the A operand before its half split (B arrives pre-split and
                                    // pre-scaled; alpha undoes both scales)
    int tiles_m, tiles_n;
    unsigned long long* prof;       // debug: per-workgroup phase timestamps (ase_hip_debug_nt_profile), else null
    int prof_clk;                   // debug: stamps 1 and 2 (main loop) in shader clocks instead of the 100 MHz clock
};


typedef __attribute__((ext_vector_type(4))) int i32x4;

// SW: operands swapped in the MFMA (D = B-fragment x A-fragment): the accumulator then holds the TRANSPOSED 32 x 32
// block - a lane owns one output ROW and 4 x 4 consecutive columns - which nt8_epilogue_rows stores straight from
// registers (no LDS transposition).
template <typename T, bool SW>
__device__ __forceinline__ f32x16 nt8_mfma(const i32x4& a, const i32x4& b, const f32x16& c) {
    typedef typename V16<T>::x8 x8;
    if constexpr (SW) return mfma16<T>(__builtin_bit_cast(x8, b), __builtin_bit_cast(x8, a), c);
    else return mfma16<T>(__builtin_bit_cast(x8, a), __builtin_bit_cast(x8, b), c);
}


#define NT8_BARRIER()                        \
    do {                                     \
        __builtin_amdgcn_sched_barrier(0);   \
        __builtin_amdgcn_s_barrier();        \
        __builtin_amdgcn_sched_barrier(0);   \
    } while (0)


// ---- row-per-lane epilogues (swapped MFMA operands): one 32 x 32 accumulator fragment of a lane = 16 outputs of ONE row, the columns
// 8 g + 4 h + q (g = e >> 2, q = e & 3, h = lane >> 5).  Two outputs at a time on the packed instructions of gfx950:
//     v_pk_fma_f32 (alpha * acc + bias) -> [bit-mask operand: v_bfe_i32 + v_and per output] -> v_cvt_pk_{bf16,f16}_f32 ->
//     v_pk_max_i16 (ReLU on the packed 16-bit patterns: sign-magnitude formats order like integers, -0 included) ->
//     [mask_out: v_pk_min_i16 with 1 = "stored value > 0", v_pk_lshlrev_b16 to its bit, v_or]
// i.e. ~3 VALU operations per output where the scalar form of rounds 2-5 (per output: fma, max, select on the runtime activation
// code, a one-sided cvt_pk + shift + or to pack, compare + select + shift-or for the mask bit) cost 8-10 - with ONE wave per SIMD
// (or two in lock-step) nothing overlaps the epilogue, and at 256 outputs per lane it was 5-6 us of a 35-us launch (round 6).
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

template <typename T> __device__ __forceinline__ uint32_t cvt_pk16(f32x2 v);
template <> __device__ __forceinline__ uint32_t cvt_pk16<bf16_t>(f32x2 v) {      // RNE
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
template <> __device__ __forceinline__ uint32_t cvt_pk16<f16_t>(f32x2 v) {       // RNE, saturating like from_f32<f16_t>
    const f32x2 c = {__builtin_amdgcn_fmed3f(v[0], -65504.f, 65504.f), __builtin_amdgcn_fmed3f(v[1], -65504.f, 65504.f)};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(c, f16x2));
}

// ---- overflow detection of the dynamic loss scale, fused into the producers (round 6; rounds 4-5 re-read every half buffer of the step,
// 0.82 GB, with ase_hip_scaler_check): a launch that was given a scale record reports an element it STORED that is non-finite or sits at
// the storage type's saturation value (f16 conversions saturate at +-65504 where autocast would produce inf; bf16 goes to inf).  On the
// packed 16-bit patterns: |x| as an unsigned integer orders like the magnitude, so one v_and + one v_pk_max_u16 per two outputs keep a
// running maximum and ONE comparison per lane at the end of the epilogue decides.
// (as an asm statement: written with __builtin_elementwise_max hipcc re-associated the running maximum of an epilogue into a tree over
//  all of a row block's outputs - 33 more live registers in a kernel whose 217 leave the 64 its co-resident neighbours run in)
__device__ __forceinline__ uint32_t ovf_fold(uint32_t run, uint32_t packed) {
    const uint32_t mag = packed & 0x7FFF7FFFu;
    asm("v_pk_max_u16 %0, %0, %1" : "+v"(run) : "v"(mag));
    return run;
}
template <typename T> __device__ __forceinline__ bool ovf_hit(uint32_t run) {
    return (run & 0xFFFFu) >= ovf_threshold<T>() || (run >> 16) >= ovf_threshold<T>();
}

// bias: the lane's 16 bias values of the fragment as 4 x f32x4 (columns 8 g + 4 h ..); w: the fragment's mask word >> 4 h (AUXK = 2);
// relu_lo: packed lower bound of v_pk_max_i16 (0 = ReLU, -32768 = none); pk[g][0 / 1]: the packed outputs q = 0, 1 / 2, 3;
// returns (MASK) the lane's 16 "stored > 0" bits at their places 8 g + 4 h + q of the fragment's 32-bit mask word
template <typename T, int AUXK, bool MASK>
__device__ __forceinline__ uint32_t rows_frag(const f32x16& c, const f32x4 (&bias)[4], uint32_t w, f32x2 al, i16x2 relu_lo, int h,
                                              uint32_t (&pk)[4][2]) {
    uint32_t mlo = 0, mhi = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
            f32x2 v = {c[g * 4 + 2 * pr], c[g * 4 + 2 * pr + 1]};
            const f32x2 b = {bias[g][2 * pr], bias[g][2 * pr + 1]};
            v = v * al + b;
            if constexpr (AUXK == 2) {
                // (as ONE vector AND: written per element - v[0] = ..., v[1] = ... - hipcc 7.2 folded the two ANDs into a v_bitop3 that
                //  multiplied element 0 into element 1; caught by the bit-mask tests of round 6)
                const i32x2 keep = {(int)__builtin_amdgcn_sbfe(w, 8 * g + 2 * pr, 1), (int)__builtin_amdgcn_sbfe(w, 8 * g + 2 * pr + 1, 1)};
                v = __builtin_bit_cast(f32x2, __builtin_bit_cast(i32x2, v) & keep);
            }
            const i16x2 o = __builtin_elementwise_max(__builtin_bit_cast(i16x2, cvt_pk16<T>(v)), relu_lo);
            pk[g][pr] = __builtin_bit_cast(uint32_t, o);
            if constexpr (MASK) {
                const i16x2 one = {1, 1};
                const u16x2 t = __builtin_bit_cast(u16x2, __builtin_elementwise_min(o, one));      // 1 where the stored value is > 0
                const u16x2 sh = {(unsigned short)((g & 1) * 8 + 2 * pr), (unsigned short)((g & 1) * 8 + 2 * pr + 1)};
                const uint32_t m = __builtin_bit_cast(uint32_t, t << sh);
                if (g < 2) mlo |= m; else mhi |= m;
            }
        }
    if constexpr (MASK) return ((((mlo | (mlo >> 16)) & 0xFFFFu) | ((mhi | (mhi >> 16)) << 16))) << (4 * h);
    else return 0u;
}

// ---- epilogue of the phased kernels with swapped MFMA operands.  acc[i][j] holds the transposed 32 x 32 block: lane
// (r = lane & 31, h = lane >> 5) owns output row i*32 + r and the columns j*32 + 8 g + 4 h + q: four runs of 4 consecutive columns.
// bias + activation + mask in registers (rows_frag), then v_permlane32_swap between the column groups (g, g + 1) of the two half-waves
// gives every lane 8 consecutive columns = ONE 16-byte store (lanes 0-31: columns 8 g .., lanes 32-63: columns 8 (g + 1) ..).
// Mask words: one 32-bit word per (row, 32-column fragment) per lane - fetched by the caller before / during the main loop (bits[]);
// the forward's mask_out word is assembled from the two half-waves' 16 bits each with one more swap.  Needs whole 64-column wave
// tiles (N % 64 == 0), a 16-bit output and act <= ASE_ACT_RELU.
// NJ: 32-column fragments of the wave tile (2: the 8-wave kernel's 128 x 64, 4: the 4-wave kernel's 128 x 128).
// NI: 32-row blocks of the wave tile (4: the 256 x 256 tiles' 128 rows per wave row, 3: the 192 x 256 tile's 96).
template <typename T, int AUXK, bool MASK, int NJ, int NI>
__device__ __forceinline__ void nt8_epilogue_rows_impl(const NTParams& p, f32x16 (&acc)[NI][NJ], int lane, int mrow0, int ncol0,
                                                       const uint32_t (&bits)[NI][NJ]) {
    const int r = lane & 31, h = lane >> 5;
    const f32x2 al = {p.alpha, p.alpha};
    const short lo = (p.act == ASE_ACT_RELU) ? (short)0 : (short)-32768;
    const i16x2 relu_lo = {lo, lo};
    uint32_t ovf = 0;
    f32x4 bias[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (p.bias) bias[j][g] = *reinterpret_cast<const f32x4*>(p.bias + ncol0 + j * 32 + 8 * g + 4 * h);
            else bias[j][g] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m = mrow0 + i * 32 + r;
        const bool row_ok = m < p.M;
        char* crow = p.C + (int64_t)m * p.ldc + (int64_t)ncol0 * 2 + h * 16;
        uint32_t mword[NJ];
        uint4 out[NJ][2];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            uint32_t pk[4][2];
            mword[j] = rows_frag<T, AUXK, MASK>(acc[i][j], bias[j], AUXK == 2 ? bits[i][j] >> (4 * h) : 0u, al, relu_lo, h, pk);
#pragma unroll
            for (int g = 0; g < 4; g += 2) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[g][0], pk[g + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[g][1], pk[g + 1][1], false, false);
                // lanes 0-31: [own g | upper's g] = columns 8 g .. 8 g + 7; lanes 32-63: [lower's g + 1 | own g + 1]
                out[j][g >> 1] = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                // (on the swapped words: they are live until the store anyway - folding pk[] in rows_frag cost 33 registers)
                ovf = ovf_fold(ovf_fold(ovf_fold(ovf_fold(ovf, s0[0]), s1[0]), s0[1]), s1[1]);
            }
        }
        if (row_ok) {                           // (one predicated block per row block: the swaps above need every lane)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int g = 0; g < 4; g += 2) *reinterpret_cast<uint4*>(crow + (j * 32 + 8 * g) * 2) = out[j][g >> 1];
        }
        if constexpr (MASK) {
#pragma unroll
            for (int jp = 0; jp < NJ; jp += 2) {
                // word j of this row = own 16 bits | the other half-wave's 16 bits
                const auto w = __builtin_amdgcn_permlane32_swap(mword[jp], mword[jp + 1], false, false);
                // after the swap: lanes 0-31 hold (own word jp bits, upper's word jp bits); lanes 32-63 (lower's word jp + 1, own)
                const uint32_t full = w[0] | w[1];
                if (row_ok) p.mask_out[(int64_t)m * p.ldmask + (ncol0 >> 5) + jp + h] = full;
            }
        }
    }
    // (rows past M are clamped copies of the last valid row, see the kernels' DMA addressing: they cannot report what a stored row does not)
    ovf_report(p.alpha_dev, ovf_hit<T>(ovf));
}

template <typename T, int AUXK, int NJ = 2, int NI = 4>
__device__ __forceinline__ void nt8_epilogue_rows(const NTParams& p, f32x16 (&acc)[NI][NJ], int lane, int mrow0, int ncol0,
                                                  const uint32_t (&bits)[NI][NJ]) {
    if (ncol0 >= p.N) return;                                   // wave-uniform: N is a multiple of the wave tile's width
    if (p.mask_out) nt8_epilogue_rows_impl<T, AUXK, true, NJ, NI>(p, acc, lane, mrow0, ncol0, bits);
    else nt8_epilogue_rows_impl<T, AUXK, false, NJ, NI>(p, acc, lane, mrow0, ncol0, bits);
}


// ---- counted waits / barriers of the phased kernels (NT and TN share the schedule) ----------------------------------
template <int UNITS> __device__ __forceinline__ void wait_dma_units() { wait_vmcnt<2 * UNITS>(); }
__device__ __forceinline__ void wait_dma_units_rt(int units) {      // wave-uniform runtime count (loop tail)
    if (units >= 4) wait_vmcnt<8>();
    else if (units == 3) wait_vmcnt<6>();
    else if (units == 2) wait_vmcnt<4>();
    else if (units == 1) wait_vmcnt<2>();
    else wait_vmcnt<0>();
}
// end of the read half of a phase: barrier, THEN retire the LDS reads (hipcc puts a vmcnt(0) in front of any LDS read that
// follows a global_load_lds without a barrier in between, so the DMA is always issued after the phase's fragment reads),
// MFMA block at raised priority
__device__ __forceinline__ void nt8_sync_in() {
    NT8_BARRIER();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
}
__device__ __forceinline__ void nt8_sync_out() {                    // end of the MFMA half
    __builtin_amdgcn_s_setprio(0);
    NT8_BARRIER();
}

// ---- host side ------------------------------------------------------------------------------------------------------
extern unsigned long long* g_nt_prof;      // tuning aid, see ase_hip_debug_nt_profile (defined in gemm.hip)
extern int g_nt_prof_clk;                  // ... stamps 1, 2 in shader clocks (ase_hip_debug_nt_profile_clock)

// Kernel choice of an NT launch (also reported by ase_hip_gemm_nt_kernel_id):
//   0:  64 x  64 tile, 4 waves   narrow heads (N <= 64): more workgroups
//   1: 128 x 128 tile, 4 waves   grids that would leave a 256 x 256 tiling with a ragged round
//   2: 256 x 256 tile, 8 waves, PHASED (16-bit storage, K in whole 128-byte steps)   192+ tiles in whole rounds
//   3: 256 x 256 tile, 8 waves, lock-step (the 4-byte storage types)
//   4:  64 x 128 tile, 4 waves / 5: 64 x 64 tile, 4 waves   small grids (M = 2048 ... 4096 rows, or N = 512): two workgroups per CU
//   6: 192 x 256 tile, 8 waves, PHASED   grids of 192 ... 255 tiles of 256 x 256 that become <= 256 tiles of 192 x 256 (the
//      discriminator's 12288 rows x 1024: 192 -> 256 workgroups)
int nt_choice(int M, int N, int K, int es, bool b16);

// one translation unit per storage type (gemm_nt_<type>.hip)
int dispatch_nt_bf16(const NTParams& p, hipStream_t s);
int dispatch_nt_f16(const NTParams& p, hipStream_t s);
int dispatch_nt_f32(const NTParams& p, hipStream_t s);
int dispatch_nt_x3(const NTParams& p, hipStream_t s);      // f32s_t: three bf16 MFMAs per product
int dispatch_nt_h3(const NTParams& p, hipStream_t s);      // f32h_t: three f16 MFMAs per product, scaled operands

}  // namespace ase_nt
