// NT, 256 x 256 tile as FOUR waves (one per SIMD, wave tile 128 x 128, 256 accumulator registers), operands HBM -> LDS by
// `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer descriptor: SGPR base + 32-bit per-lane offset + SGPR K-tile offset, M0 =
// destination), software-pipelined HALF a K-tile deep.  16-bit storage, gfx950.  C = mask(act(alpha A B^T + bias)).
//
// LAB VARIANT (round 6), linked into libase_hip_lab.so; ASE_NT4V=<variant bits> dispatches it from lab_nt_bf16.hip.
//
// Why another 4-wave kernel.  Rounds 2-3 built two (gemm_nt4_variant.inc: LDS-DMA through global_load_lds, gemm_nt4r_variant.hip:
// register staging) and neither beat the 8-wave phased kernel; the vendor library's kernel for the same shape IS a 4-wave 128 x 128
// kernel with both operands direct-to-LDS (profiles/r04_rocblas_kernel_info.txt) and runs 20-28 % ahead.  Its disassembly (llvm-objdump of
// the hipBLASLt code object, round 6) shows what the lab variants did not have:
//   * the DMA is `buffer_load_dwordx4 v, s[rsrc], soff offen lds` - one SGPR descriptor, a 32-bit offset register per piece, no per-piece
//     64-bit address arithmetic; M0 moves by `s_add_u32 m0, ...` between pieces;
//   * the loop is pipelined by HALF K-tiles: the MFMAs of k-half 0 run while k-half 1 of the same K-tile is read from LDS, the MFMAs of
//     k-half 1 while k-half 0 of the NEXT K-tile is read; the DMA of K-tile t + 2 refills the buffer of K-tile t as soon as a barrier says
//     every wave has its fragments of that operand in registers (A first, then B);
//   * at most ONE memory instruction between two MFMAs, DMA pieces at least two MFMAs apart;
//   * two copies of the loop, chosen by the SIMD the wave sits on (s_getreg HW_ID), whose memory instructions are one MFMA apart - the
//     waves of a workgroup do not present their DMA pieces to the texture path in the same cycle.
// This file is that schedule on OUR K-tile image (128-byte rows, 16-byte chunk c of row r at slot c ^ ((r >> 1) & 7), DMA piece = 8 rows x
// 128 B = one wave instruction) with 32 x 32 x 16 MFMAs on swapped operands, so that the product's row-per-lane epilogue
// (nt8_epilogue_rows) applies unchanged.
//
// K-tile t (buffer t & 1 of two 64-KiB images [A: 256 rows][B: 256 rows]), slot n = the gap behind MFMA n of its 64 MFMAs
// (k-step ks = n / 16, (i, j) = 32 x 32 block of the wave tile):
//     n  0..14 even   A fragments of ks 2, 3 of K-tile t              (8 ds_read_b128)
//     n 16 / 17       lgkmcnt(0) / barrier 1: every wave holds its A fragments -> A half of the buffer is free
//     n 18..25        B fragments of ks 2, 3                          (8 reads)
//     n 18..32 even   DMA A pieces 0..7 of K-tile t + 2               (odd-SIMD waves: n 19..33 odd)
//     n 26 / 27       lgkmcnt(0) / barrier 2: B half of the buffer is free
//     n 34..48 even   DMA B pieces 0..7 of K-tile t + 2               (odd-SIMD waves: 35..49 odd)
//     n 36 / 37       vmcnt(17) / barrier 3: A of K-tile t + 1 has landed for every wave
//     n 38..45        A fragments of ks 0, 1 of K-tile t + 1          (8 reads)
//     n 50 / 51       vmcnt(16) / barrier 4: B of K-tile t + 1 has landed
//     n 52..59        B fragments of ks 0, 1 of K-tile t + 1          (8 reads)
//     n 63            lgkmcnt(0)
// V bits (timing ablations give wrong results): 1 = DMA by the global_load_lds builtin (64-bit per-lane addresses, as the product's kernels
// issue it), 2 = NO stagger between even / odd SIMDs, 4 = no DMA, 8 = no fragment reads, 16 = shader-clock stamps of the loop.
#include "../../ase_amd/csrc/gemm_nt.h"
#include <stdlib.h>
#include <utility>

using namespace ase_nt;

namespace {

struct NT4V {
    uint32_t voffA[8], voffB[8];       // per-lane byte offset of DMA piece q from the operand's base: row * ld + swizzled 16-byte chunk
    // per-lane LDS byte address [k-step] of the wave's first A / B fragment row block: k-steps 2, 3 in the buffer of the CURRENT K-tile,
    // k-steps 0, 1 in the buffer of the NEXT one; all eight flip buffers (^ 65536) at the end of every K-tile
    uint32_t rdA[4], rdB[4];
    uint32_t dstA, dstB;               // wave-uniform LDS byte address of this wave's piece 0 in the buffer the DMA refills (piece q: + q * 4096)
    const char* gA[8];                 // V & 1: per-lane 64-bit sources
    const char* gB[8];
};

template <int OFF> __device__ __forceinline__ void v_read(i32x4& f, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF) : "memory");
}
// retire LDS reads; naming the fragment registers keeps every MFMA that uses them behind the wait
__device__ __forceinline__ void v_retire8(i32x4 (&f0)[4], i32x4 (&f1)[4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f0[0]), "+v"(f0[1]), "+v"(f0[2]), "+v"(f0[3]), "+v"(f1[0]), "+v"(f1[1]), "+v"(f1[2]), "+v"(f1[3])
                 :
                 : "memory");
}
// one DMA piece: M0 = wave-uniform LDS destination, lane l lands at M0 + 16 l
template <int IMM> __device__ __forceinline__ void v_dma(uint32_t lds_s, uint32_t voff, i32x4 rs, uint32_t soff) {
    asm volatile("s_add_u32 m0, %0, %1\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                 :
                 : "s"(lds_s), "n"(IMM), "v"(voff), "s"(rs), "s"(soff)
                 : "memory", "scc");
}

// DMA piece q of operand A / B into the buffer L.dst* points at, K-tile byte offset soff
template <int V, int q, bool isB>
__device__ __forceinline__ void v_piece(const NT4V& L, i32x4 rsA, i32x4 rsB, uint32_t soff) {
    if constexpr (V & 1) {
        const uint32_t d = (isB ? L.dstB : L.dstA) + q * 4096;
        __builtin_amdgcn_global_load_lds((gptr_t*)((isB ? L.gB[q] : L.gA[q]) + soff), (lptr_t*)(uintptr_t)d, 16, 0, 0);
    } else {
        v_dma<q * 4096>(isB ? L.dstB : L.dstA, isB ? L.voffB[q] : L.voffA[q], isB ? rsB : rsA, soff);
    }
}
template <int V, int... Q>
__device__ __forceinline__ void v_tile_dma(const NT4V& L, i32x4 rsA, i32x4 rsB, uint32_t soff, std::integer_sequence<int, Q...>) {
    (v_piece<V, Q, false>(L, rsA, rsB, soff), ...);
    (v_piece<V, Q, true>(L, rsA, rsB, soff), ...);
}

// one K-tile = 64 MFMAs with the memory instructions of the table above in the gaps.
// DMA: K-tile t + 2 exists (soff = its byte offset inside a row); NEXT: K-tile t + 1 exists (its fragments are read, its landing is waited for)
template <typename T, int V, bool ODD, bool DMA, bool NEXT>
struct Ktile {
    f32x16 (&acc)[4][4];
    i32x4 (&FA)[4][4];                 // [ks][i]
    i32x4 (&FB)[4][4];                 // [ks][j]
    NT4V& L;
    const i32x4 rsA, rsB;
    const uint32_t soff;

    template <int q, bool isB> __device__ __forceinline__ void dma() const {
        if constexpr (DMA && !(V & 4)) v_piece<V, q, isB>(L, rsA, rsB, soff);
    }
    template <int n> __device__ __forceinline__ void slot() const {
        constexpr bool RD = !(V & 8);
        constexpr int sh = (ODD && !(V & 2)) ? 1 : 0;
        __builtin_amdgcn_sched_barrier(0);
        // ---- waits and barriers (same slots for every wave)
        if constexpr (n == 16) { if constexpr (RD) { v_retire8(FA[2], FA[3]); } }
        if constexpr (n == 17) NT8_BARRIER();
        if constexpr (n == 26) { if constexpr (RD) { v_retire8(FB[2], FB[3]); } }
        if constexpr (n == 27) NT8_BARRIER();
        if constexpr (n == 36 && NEXT) {
            if constexpr (DMA) wait_vmcnt<17>(); else wait_vmcnt<8>();
        }
        if constexpr (n == 37 && NEXT) NT8_BARRIER();
        if constexpr (n == 50 && NEXT) {
            if constexpr (DMA) wait_vmcnt<16>(); else wait_vmcnt<0>();
        }
        if constexpr (n == 51 && NEXT) NT8_BARRIER();
        // ---- fragment reads
        if constexpr (RD) {
            if constexpr (n <= 14 && (n & 1) == 0) {                  // A, ks 2 / 3 of this K-tile
                constexpr int k = n / 2, ks = 2 + k / 4, i = k % 4;
                v_read<i * 4096>(FA[ks][i], L.rdA[ks]);
            }
            if constexpr (n >= 18 && n <= 25) {                       // B, ks 2 / 3
                constexpr int k = n - 18, ks = 2 + k / 4, j = k % 4;
                v_read<j * 4096>(FB[ks][j], L.rdB[ks]);
            }
            if constexpr (NEXT && n >= 38 && n <= 45) {               // A, ks 0 / 1 of the next K-tile
                constexpr int k = n - 38, ks = k / 4, i = k % 4;
                v_read<i * 4096>(FA[ks][i], L.rdA[ks]);
            }
            if constexpr (NEXT && n >= 52 && n <= 59) {               // B, ks 0 / 1 of the next K-tile
                constexpr int k = n - 52, ks = k / 4, j = k % 4;
                v_read<j * 4096>(FB[ks][j], L.rdB[ks]);
            }
            if constexpr (NEXT && n == 63) {
                v_retire8(FA[0], FA[1]);
                v_retire8(FB[0], FB[1]);
            }
        }
        // ---- every read address and the DMA destination flip buffers for the next K-tile (behind their last use)
        if constexpr (n == 28) { L.rdA[2] ^= 65536u; L.rdA[3] ^= 65536u; }
        if constexpr (n == 30) { L.rdB[2] ^= 65536u; L.rdB[3] ^= 65536u; }
        if constexpr (n == 60) { L.rdA[0] ^= 65536u; L.rdA[1] ^= 65536u; }
        if constexpr (n == 61) { L.rdB[0] ^= 65536u; L.rdB[1] ^= 65536u; }
        if constexpr (n == 62) { L.dstA ^= 65536u; L.dstB ^= 65536u; }
        // ---- DMA of K-tile t + 2 into this K-tile's buffer
        if constexpr (n >= 18 + sh && n <= 32 + sh && ((n - 18 - sh) & 1) == 0) dma<(n - 18 - sh) / 2, false>();
        if constexpr (n >= 34 + sh && n <= 48 + sh && ((n - 34 - sh) & 1) == 0) dma<(n - 34 - sh) / 2, true>();
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int n> __device__ __forceinline__ void step() const {
        constexpr int ks = n / 16, i = (n % 16) % 4, j = (n % 16) / 4;
        acc[i][j] = nt8_mfma<T, true>(FA[ks][i], FB[ks][j], acc[i][j]);
        slot<n>();
    }
    template <int... N> __device__ __forceinline__ void run(std::integer_sequence<int, N...>) const { (step<N>(), ...); }
};

// every K-tile but the last
template <typename T, int V, bool ODD>
__device__ __forceinline__ void v_loop(int nk, f32x16 (&acc)[4][4], i32x4 (&FA)[4][4], i32x4 (&FB)[4][4], NT4V& L, i32x4 rsA, i32x4 rsB) {
    uint32_t soff = 256;
    for (int t = 0; t + 2 < nk; ++t) {
        Ktile<T, V, ODD, true, true> k{acc, FA, FB, L, rsA, rsB, soff};
        k.run(std::make_integer_sequence<int, 64>{});
        soff += 128;
    }
    if (nk > 1) {
        Ktile<T, V, ODD, false, true> k{acc, FA, FB, L, rsA, rsB, 0u};
        k.run(std::make_integer_sequence<int, 64>{});
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void gemm_nt4v_kernel(NTParams p) {
    static_assert(sizeof(T) == 2, "16-bit storage only");
    constexpr int RB = 128, BM = 256, BK = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int bm0 = (tile / p.tiles_n) * BM, bn0 = (tile % p.tiles_n) * 256;

    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 0] = wall_clock64();
    // whole operands as raw buffers (stride 0, byte extent): rows past M / N are CLAMPED to the last row in the per-lane offsets (their
    // products only reach outputs that are never stored)
    i32x4 rsA, rsB;
    {
        const uint64_t a = (uint64_t)(uintptr_t)p.A, b = (uint64_t)(uintptr_t)p.B;
        rsA = i32x4{(int)(uint32_t)a, (int)(uint32_t)(a >> 32), (int)0x7FFFFFFF, 0x00020000};
        rsB = i32x4{(int)(uint32_t)b, (int)(uint32_t)(b >> 32), (int)0x7FFFFFFF, 0x00020000};
    }
    NT4V L;
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;          // (the one LDS object of the kernel: offset 0, so ^ 65536 flips buffers)
    {
        const int lr = lane >> 3, slot = lane & 7;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = (q * 4 + wid) * 8 + lr;                       // tile row of this lane in piece q
            const int64_t ga = min(bm0 + r, p.M - 1), gb = min(bn0 + r, p.N - 1);
            const uint32_t ch = (uint32_t)((slot ^ lds_swz<RB>(r)) << 4);
            L.voffA[q] = (uint32_t)(ga * p.lda) + ch;
            L.voffB[q] = (uint32_t)(gb * p.ldb) + ch;
            if constexpr (V & 1) {
                L.gA[q] = p.A + ga * p.lda + ch;
                L.gB[q] = p.B + gb * p.ldb + ch;
            }
        }
        L.dstA = __builtin_amdgcn_readfirstlane(lds0 + wid * 1024);
        L.dstB = __builtin_amdgcn_readfirstlane(lds0 + 32768 + wid * 1024);
        const int r = lane & 31, h = lane >> 5, sw = lds_swz<RB>(r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t ro = r * RB + (((ks * 2 + h) ^ sw) << 4);
            L.rdA[ks] = lds0 + wr * 128 * RB + ro;             // buffer 0 (K-tile 0)
            L.rdB[ks] = lds0 + 32768 + wc * 128 * RB + ro;
        }
    }

    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = p.K / BK;
    // (a descriptor SGPR written by a VALU readfirstlane needs 5 wait states before a buffer instruction reads it)
    asm volatile("s_nop 4" ::: "memory");
    // prologue: K-tiles 0 and 1 into the two buffers; the fragments of k-steps 0, 1 of K-tile 0 into registers
    v_tile_dma<(V & ~4)>(L, rsA, rsB, 0u, std::make_integer_sequence<int, 8>{});
    L.dstA ^= 65536u; L.dstB ^= 65536u;
    if (nk > 1) v_tile_dma<(V & ~4)>(L, rsA, rsB, 128u, std::make_integer_sequence<int, 8>{});
    // mask words of the wave tile (data-gradient launches: 128 rows x 4 words) as eight 4-byte DMA pieces BEHIND the prologue's K-tiles
    // into 2 KiB of LDS per wave past the ring: no registers, no wait before the epilogue (lane (r, h) fetches words h and h + 2 of
    // row 32 i + r).  Older than every DMA piece of the loop, they never disturb its counted waits.
    const bool masked = p.aux_mode == ASE_AUX_RELU_BITS && bn0 + wc * 128 < p.N;
    if (masked) {
        char* mlds = smem + 2 * 65536 + wid * 2048;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = bm0 + wr * 128 + i * 32 + (lane & 31);
            const int ma = (m >= p.aux_split) ? m - p.aux_delta : m;
            const uint32_t* w = reinterpret_cast<const uint32_t*>(p.aux + (int64_t)min(ma, p.M - 1) * p.ldaux) +
                                ((bn0 + wc * 128) >> 5) + (lane >> 5);
            __builtin_amdgcn_global_load_lds((gptr_t*)w, (lptr_t*)(mlds + (i * 2) * 256), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t*)(w + 2), (lptr_t*)(mlds + (i * 2 + 1) * 256), 4, 0, 0);
        }
    }
    if (nk > 1) {
        if (masked) wait_vmcnt<24>(); else wait_vmcnt<16>();
    } else {
        wait_vmcnt<0>();
    }
    L.dstA ^= 65536u; L.dstB ^= 65536u;         // the loop's first DMA (K-tile 2) refills buffer 0
    NT8_BARRIER();
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 1] = wall_clock64();
    i32x4 FA[4][4], FB[4][4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        v_read<0>(FA[ks][0], L.rdA[ks]); v_read<4096>(FA[ks][1], L.rdA[ks]);
        v_read<8192>(FA[ks][2], L.rdA[ks]); v_read<12288>(FA[ks][3], L.rdA[ks]);
        v_read<0>(FB[ks][0], L.rdB[ks]); v_read<4096>(FB[ks][1], L.rdB[ks]);
        v_read<8192>(FB[ks][2], L.rdB[ks]); v_read<12288>(FB[ks][3], L.rdB[ks]);
    }
    v_retire8(FA[0], FA[1]);
    v_retire8(FB[0], FB[1]);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {             // from now on k-steps 0, 1 are read from the NEXT K-tile's buffer
        L.rdA[ks] ^= 65536u;
        L.rdB[ks] ^= 65536u;
    }
    if constexpr (V & 8) {          // ablation without reads: the fragments must still be defined values
#pragma unroll
        for (int ks = 2; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) { FA[ks][i] = FA[0][i]; FB[ks][i] = FB[0][i]; }
    }

    long long clk0 = 0, clk1 = 0;
    if constexpr (V & 16) clk0 = clock64();
    v_loop<T, V, false>(nk, acc, FA, FB, L, rsA, rsB);
    {
        Ktile<T, V, false, false, false> k{acc, FA, FB, L, rsA, rsB, 0u};
        k.run(std::make_integer_sequence<int, 64>{});
    }
    if constexpr (V & 16) clk1 = clock64();
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 2] = wall_clock64();

    uint32_t row_bits[4][4];
    if (masked) {
        const uint32_t* mw = reinterpret_cast<const uint32_t*>(smem + 2 * 65536 + wid * 2048);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) row_bits[i][j] = mw[(i * 2 + (j >> 1)) * 64 + (j & 1) * 32 + (lane & 31)];
        nt8_epilogue_rows<T, 2, 4>(p, acc, lane, bm0 + wr * 128, bn0 + wc * 128, row_bits);
    } else
        nt8_epilogue_rows<T, 0, 4>(p, acc, lane, bm0 + wr * 128, bn0 + wc * 128, row_bits);
    if (p.prof) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) p.prof[blockIdx.x * 4 + 3] = wall_clock64();
        if constexpr (V & 16) {          // "epilogue+drain" column of the lab print x 1000 = shader clocks of the main loop
            if (tid == 0) p.prof[blockIdx.x * 4 + 3] = p.prof[blockIdx.x * 4 + 2] + (unsigned long long)(clk1 - clk0) / 10;
        }
    }
}

}  // namespace

namespace ase_nt {

template <typename T> int launch_nt4v(const NTParams& p0, unsigned long long* prof, int variant, hipStream_t stream) {
    constexpr int lds = 2 * 512 * 128 + 4 * 2048;          // ring + 2 KiB of mask words per wave
    typedef void (*kern_t)(NTParams);
    kern_t kern = nullptr;
    switch (variant) {
        case 2: kern = gemm_nt4v_kernel<T, 2>; break;      // the kernel
        case 3: kern = gemm_nt4v_kernel<T, 3>; break;      // ... DMA by the global_load_lds builtin
        case 6: kern = gemm_nt4v_kernel<T, 6>; break;      // ablation: no DMA
        case 10: kern = gemm_nt4v_kernel<T, 10>; break;    // ablation: no fragment reads
        case 14: kern = gemm_nt4v_kernel<T, 14>; break;    // ablation: MFMAs + barriers only
        case 18: kern = gemm_nt4v_kernel<T, 18>; break;    // one copy + shader-clock stamps
        case 30: kern = gemm_nt4v_kernel<T, 30>; break;    // MFMAs only + stamps
        default: ase_set_error("gemm_nt4v: unknown variant %d", variant); return ASE_EINVAL;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) {
        ase_set_error("gemm_nt4v: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return ASE_ELAUNCH;
    }
    NTParams p = p0;
    p.prof = prof;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    ASE_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n), dim3(256), lds, stream, p);
    ASE_CHECK_LAUNCH("gemm_nt4v");
    return ASE_OK;
}

template int launch_nt4v<bf16_t>(const NTParams&, unsigned long long*, int, hipStream_t);

}  // namespace ase_nt
