// NT, 256 x 256 tile as FOUR waves with REGISTER-staged operands (16-bit storage, gfx950):  C = mask(act(alpha A B^T + bias)).
//
// Why a second 256 x 256 kernel.  The 8-wave phased kernel (gemm_nt8_kernel) is bound by LDS data movement: its 128 x 64 wave
// tiles read 192 KB of fragments per K-tile and CU and the LDS-DMA writes 64 KB - at 128 B/clk as long as the K-tile's MFMAs
// (2048 clk) - and the ablations showed those costs do not hide under the partner wave's matrix work.  128 x 128 wave tiles
// (one wave per SIMD, 256 accumulator registers) cut the fragment reads to 128 KB; built with LDS-DMA staging in round 2 the
// reads did hide, but every global_load_lds stalled its wave ~175 cycles with no partner wave to cover it (8192^3: 1143 us; the
// same kernel WITHOUT its DMA: 765 us - the vendor library's 781).  So the staging here is the classic one: buffer_load_dwordx4
// into registers four phases ahead, ds_write_b128 into the ring later - both cost a few issue cycles between MFMAs.
//
// K-tile image (64 KB, two of them): [A: 256 rows x 128 B][B: 256 rows x 128 B], 16-byte chunk c of row r at slot
// c ^ ((r >> 1) & 7).  A "unit" is a quarter of a K-tile: A0 / B0 = rows 0-63 of every wave's 128 A / B rows, B1 / A1 = rows
// 64-127 (order of first use); unit u = 4 t + kind.  Per wave and unit: 4 pieces of 1 KiB (8 rows x 128 B, lane = row x chunk).
// One K-tile = four phases of 16 MFMAs over the quadrants of the wave tile
//     q0: A01 x B01   q1: A01 x B23   q2: A23 x B23   q3: A23 x B01            (A01 = 32-row blocks 0, 1 of the wave's rows ...)
// and phase p = 4 t + q also
//     reads   the fragment set of unit p + 2   (q0: B23(t)  q1: A23(t)  q2: A01(t+1)  q3: B01(t+1); 8 ds_read_b128)
//     writes  unit p + 4 from its staging registers into the other K-tile image (4 ds_write_b128; loaded in phase p - 4)
//     loads   unit p + 8 into the same staging registers (4 buffer_load_dwordx4; rows past M / N read as zero)
// one LDS or memory instruction behind each MFMA.  Hazards: unit u is written in phase u - 4 and read in phase u - 2, and its
// ring slot was last read in phase u - 10 - a barrier at the top of every EVEN phase orders both (two per K-tile).  Every
// phase ends with lgkmcnt(0) (its reads feed the next phase, its writes must be done before the next barrier); vmcnt is
// counted by the compiler (the loads are builtins, only the LDS instructions are inline asm so that nothing reorders them).
// Registers: 256 accumulators + 4 fragment sets (128) + 4 staging sets (64) + addresses.
//
// LAB VARIANT, NOT part of the product build (scripts/lab/Makefile links it into libase_hip_lab.so; ASE_NT4R=1 dispatches it,
// ASE_NT4R_ABL selects a timing ablation).  MEASURED (round 3, scripts/lab/r3i.sh / r3k.sh, profiles/r03_lab_nt4r.log):
// correct on every shape and in the product's whole GPU test-suite (193 tests with it dispatched), and NOT faster than the
// 8-wave phased kernel: main loop of the 16384 x 1024 x 1024 layer 24.9 us (8-wave: 25.4), 32768 x 1024 x 1024 69.1 vs 69.6 us
// per launch, 8192^3 1045 vs 962 us, launches with a mask operand 1-2 us slower (the mask words are ordinary loads here).
// Ablations on the 16384 x 1024 x 1024 layer (loop us): full 24.9 | no global loads 23.1 | no loads, no LDS writes 21.2 | no
// fragment reads 20.9 | MFMAs + barriers only 17.8 - the costs are ADDITIVE here too (~18 cycles of matrix-pipe time per
// memory instruction issued between two MFMAs of the only wave of a SIMD).  Shader clocks of the loop (s_memtime against the
// 100 MHz s_memrealtime): 39.85 k cycles in 25.0 us = 1.59 GHz for the full kernel, 34.3 k in 17.95 us = 1.91 GHz MFMA-only,
// 8192^3 1.73 / 2.05 GHz: the chip does not hold its 2.4 GHz boost under dense MFMA load on random operands.  In CYCLES the
// loop keeps the matrix pipe busy 82 % of the time (2048 of 2490 per K-tile); the rest of the gap to the 2.5 PF dense peak is
// frequency.  That - not the staging mechanism - is what both 256 x 256 kernels are up against.
#include "../../ase_amd/csrc/gemm_nt.h"
#include <stdlib.h>

using namespace ase_nt;

namespace {

constexpr int kRsrcFlags = 0x00020000;      // raw buffer, 32-bit data format (gfx9 / CDNA resource word 3)

struct NT4Lane {
    uint32_t voffA, voffB;     // per-lane byte offset inside a piece's 8 rows: row (wid * 8 + lane / 8), swizzled source chunk
    uint32_t wr[2];            // per-lane LDS byte address of this lane's slot in a piece, K-tile image 0 / 1
    uint32_t adA[4], adB[4];   // per-lane LDS byte address (image 0) of the wave's A / B fragment rows, per k-step
};

template <int OFF> __device__ __forceinline__ void nt4_read1(i32x4& f, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ void nt4_write1(uint32_t addr, const i32x4& v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
// retires the phase's LDS traffic; naming the fragment registers keeps every MFMA that uses them behind the wait
__device__ __forceinline__ void nt4_retire(i32x4 (&f)[2][4]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[0][2]), "+v"(f[0][3]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[1][2]),
                   "+v"(f[1][3])
                 :
                 : "memory");
}

// row offset (inside the 256-row operand tile) of piece g of a unit kind: the two wave-row groups x the two 32-row blocks
__device__ constexpr int nt4_piece_row(int kind, int g) { return (g >> 1) * 128 + ((kind >= 2) ? 64 : 0) + (g & 1) * 32; }
__device__ constexpr bool nt4_kind_is_b(int kind) { return kind == 1 || kind == 2; }

// one phase: c{ij} += a[i] x b[j] over the 4 k-steps.
//   SUB: the fragment set read for the next phase = 64-row half SUB of the wave's A or B rows, at LDS addresses rd[ks]
//   KIND: the unit kind written (K-tile image `wimg`, if wr_live) and loaded (K-tile `ltile`, if ld_live) - staging set st
// ABL (lab builds only, timing ablations with wrong results): 1 no global loads, 2 no LDS writes, 4 no fragment reads
template <typename T, int SUB, int KIND, int ABL>
__device__ __forceinline__ void nt4_phase(f32x16& c00, f32x16& c10, f32x16& c01, f32x16& c11, const i32x4 (&a)[2][4],
                                          const i32x4 (&b)[2][4], i32x4 (&nx)[2][4], const uint32_t (&rd)[4], bool rd_live,
                                          const NT4Lane& L, i32x4 (&st)[4], uint32_t wbase, bool wr_live,
                                          __amdgpu_buffer_rsrc_t rs, uint32_t voff, int64_t ld, int ltile, bool ld_live) {
    constexpr bool isB = nt4_kind_is_b(KIND);
    auto slot = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (n < 8) {                                   // fragment reads first: the next phase starts with them
            if (!(ABL & 4) && rd_live) {
                constexpr int ks = n & 3, off = SUB * 8192 + (n >> 2) * 4096;
                nt4_read1<off>(nx[n >> 2][ks], rd[ks]);
            }
        } else if constexpr (n < 12) {
            constexpr int g = n - 8;
            if (!(ABL & 2) && wr_live) nt4_write1<(isB ? 32768 : 0) + nt4_piece_row(KIND, g) * 128>(wbase, st[g]);
        } else {
            constexpr int g = n - 12;
            if (!(ABL & 1) && ld_live) {
                const uint32_t soff = (uint32_t)(nt4_piece_row(KIND, g) * ld) + (uint32_t)ltile * 128u;
                st[g] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0));
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#define NT4_SLOT(n) slot(std::integral_constant<int, n>{})
    {
        typedef typename V16<T>::x8 x8;
        (void)sizeof(x8);
        c00 = nt8_mfma<T, true>(a[0][0], b[0][0], c00); NT4_SLOT(0);
        c10 = nt8_mfma<T, true>(a[1][0], b[0][0], c10); NT4_SLOT(1);
        c01 = nt8_mfma<T, true>(a[0][0], b[1][0], c01); NT4_SLOT(2);
        c11 = nt8_mfma<T, true>(a[1][0], b[1][0], c11); NT4_SLOT(3);
        c00 = nt8_mfma<T, true>(a[0][1], b[0][1], c00); NT4_SLOT(4);
        c10 = nt8_mfma<T, true>(a[1][1], b[0][1], c10); NT4_SLOT(5);
        c01 = nt8_mfma<T, true>(a[0][1], b[1][1], c01); NT4_SLOT(6);
        c11 = nt8_mfma<T, true>(a[1][1], b[1][1], c11); NT4_SLOT(7);
        c00 = nt8_mfma<T, true>(a[0][2], b[0][2], c00); NT4_SLOT(8);
        c10 = nt8_mfma<T, true>(a[1][2], b[0][2], c10); NT4_SLOT(9);
        c01 = nt8_mfma<T, true>(a[0][2], b[1][2], c01); NT4_SLOT(10);
        c11 = nt8_mfma<T, true>(a[1][2], b[1][2], c11); NT4_SLOT(11);
        c00 = nt8_mfma<T, true>(a[0][3], b[0][3], c00); NT4_SLOT(12);
        c10 = nt8_mfma<T, true>(a[1][3], b[0][3], c10); NT4_SLOT(13);
        c01 = nt8_mfma<T, true>(a[0][3], b[1][3], c01); NT4_SLOT(14);
        c11 = nt8_mfma<T, true>(a[1][3], b[1][3], c11); NT4_SLOT(15);
    }
#undef NT4_SLOT
    __builtin_amdgcn_sched_barrier(0);
    nt4_retire(nx);
    __builtin_amdgcn_sched_barrier(0);
}

// one K-tile.  On entry X = A01(t), Bp = B01(t) are in registers; on exit X = A01(t + 1), Bq = B01(t + 1).
// TAIL: one of the last two K-tiles (reads / writes / loads of K-tiles past the end are switched off).
template <typename T, bool TAIL, int ABL>
__device__ __forceinline__ void nt4_ktile(int t, int nk, const NT4Lane& L, f32x16 (&acc)[4][4], i32x4 (&X)[2][4],
                                          i32x4 (&Y)[2][4], i32x4 (&Bp)[2][4], i32x4 (&Bq)[2][4], i32x4 (&S)[4][4],
                                          __amdgpu_buffer_rsrc_t rsA, __amdgpu_buffer_rsrc_t rsB, int64_t lda, int64_t ldb) {
    const uint32_t cur = (t & 1) * 65536, nxt = 65536 - cur;
    const bool more = !TAIL || t + 1 < nk, ld = !TAIL || t + 2 < nk;
    const uint32_t wb = L.wr[(t + 1) & 1];
    uint32_t ra[4], rb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        ra[ks] = L.adA[ks] + cur;
        rb[ks] = L.adB[ks] + cur;
    }
    NT8_BARRIER();
    nt4_phase<T, 1, 0, ABL>(acc[0][0], acc[1][0], acc[0][1], acc[1][1], X, Bp, Bq, rb, true, L, S[0], wb, more, rsA, L.voffA, lda,
                       t + 2, ld);                                                   // reads B23(t); writes A0(t+1); loads A0(t+2)
    nt4_phase<T, 1, 1, ABL>(acc[0][2], acc[1][2], acc[0][3], acc[1][3], X, Bq, Y, ra, true, L, S[1], wb, more, rsB, L.voffB, ldb,
                       t + 2, ld);                                                   // reads A23(t); writes B0(t+1); loads B0(t+2)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        ra[ks] = L.adA[ks] + nxt;
        rb[ks] = L.adB[ks] + nxt;
    }
    NT8_BARRIER();
    nt4_phase<T, 0, 2, ABL>(acc[2][2], acc[3][2], acc[2][3], acc[3][3], Y, Bq, X, ra, more, L, S[2], wb, more, rsB, L.voffB, ldb,
                       t + 2, ld);                                                   // reads A01(t+1); writes B1(t+1); loads B1(t+2)
    nt4_phase<T, 0, 3, ABL>(acc[2][0], acc[3][0], acc[2][1], acc[3][1], Y, Bp, Bq, rb, more, L, S[3], wb, more, rsA, L.voffA, lda,
                       t + 2, ld);                                                   // reads B01(t+1); writes A1(t+1); loads A1(t+2)
}

template <typename T, int ABL>
__global__ __launch_bounds__(256) void gemm_nt4r_kernel(NTParams p) {
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
    // the operand tiles as buffer resources: 256 rows from the tile's first row, rows past the matrix read as zeros
    const int rowsA = min(BM, p.M - bm0), rowsB = min(256, p.N - bn0);
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.A + (int64_t)bm0 * p.lda), 0, (int)(rowsA * p.lda), kRsrcFlags);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.B + (int64_t)bn0 * p.ldb), 0, (int)(rowsB * p.ldb), kRsrcFlags);
    NT4Lane L;
    {
        const int rl = wid * 8 + (lane >> 3), slot = lane & 7;
        const int sw = (rl >> 1) & 7;                    // = lds_swz<128>(row) for every piece: their row offsets are multiples of 32
        L.voffA = (uint32_t)(rl * p.lda) + ((slot ^ sw) << 4);
        L.voffB = (uint32_t)(rl * p.ldb) + ((slot ^ sw) << 4);
        const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
        L.wr[0] = lds0 + wid * 1024 + lane * 16;
        L.wr[1] = L.wr[0] + 65536;
        const int r = lane & 31, h = lane >> 5, swr = lds_swz<RB>(r);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t ro = r * RB + (((ks * 2 + h) ^ swr) << 4);
            L.adA[ks] = lds0 + wr * 128 * RB + ro;
            L.adB[ks] = lds0 + BM * RB + wc * 128 * RB + ro;
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
    // prologue: K-tile 0 through the staging registers into image 0, K-tile 1 into the staging registers (phase p writes
    // unit p + 4), then X = A01(0), P = B01(0)
    i32x4 S[4][4];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int kind = 0; kind < 4; ++kind)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bool isB = nt4_kind_is_b(kind);
                const uint32_t soff = (uint32_t)(nt4_piece_row(kind, g) * (isB ? p.ldb : p.lda)) + (uint32_t)t * 128u;
                S[kind][g] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(isB ? rsB : rsA, isB ? L.voffB : L.voffA, soff, 0));
            }
    };
    load_tile(0);
    nt4_write1<0 + nt4_piece_row(0, 0) * 128>(L.wr[0], S[0][0]);
    nt4_write1<0 + nt4_piece_row(0, 1) * 128>(L.wr[0], S[0][1]);
    nt4_write1<0 + nt4_piece_row(0, 2) * 128>(L.wr[0], S[0][2]);
    nt4_write1<0 + nt4_piece_row(0, 3) * 128>(L.wr[0], S[0][3]);
    nt4_write1<32768 + nt4_piece_row(1, 0) * 128>(L.wr[0], S[1][0]);
    nt4_write1<32768 + nt4_piece_row(1, 1) * 128>(L.wr[0], S[1][1]);
    nt4_write1<32768 + nt4_piece_row(1, 2) * 128>(L.wr[0], S[1][2]);
    nt4_write1<32768 + nt4_piece_row(1, 3) * 128>(L.wr[0], S[1][3]);
    nt4_write1<32768 + nt4_piece_row(2, 0) * 128>(L.wr[0], S[2][0]);
    nt4_write1<32768 + nt4_piece_row(2, 1) * 128>(L.wr[0], S[2][1]);
    nt4_write1<32768 + nt4_piece_row(2, 2) * 128>(L.wr[0], S[2][2]);
    nt4_write1<32768 + nt4_piece_row(2, 3) * 128>(L.wr[0], S[2][3]);
    nt4_write1<0 + nt4_piece_row(3, 0) * 128>(L.wr[0], S[3][0]);
    nt4_write1<0 + nt4_piece_row(3, 1) * 128>(L.wr[0], S[3][1]);
    nt4_write1<0 + nt4_piece_row(3, 2) * 128>(L.wr[0], S[3][2]);
    nt4_write1<0 + nt4_piece_row(3, 3) * 128>(L.wr[0], S[3][3]);
    __builtin_amdgcn_sched_barrier(0);
    if (nk > 1) load_tile(1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    NT8_BARRIER();
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 1] = wall_clock64();
    i32x4 X[2][4], Y[2][4], P[2][4], Q[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        nt4_read1<0>(X[0][ks], L.adA[ks]);
        nt4_read1<4096>(X[1][ks], L.adA[ks]);
        nt4_read1<0>(P[0][ks], L.adB[ks]);
        nt4_read1<4096>(P[1][ks], L.adB[ks]);
    }
    nt4_retire(X);
    nt4_retire(P);

    long long clk0 = 0;
    if constexpr (ABL & 8) clk0 = clock64();                 // lab: shader clocks of the main loop (actual frequency under load)
    int t = 0;
    for (; t + 3 < nk; t += 2) {
        nt4_ktile<T, false, ABL>(t, nk, L, acc, X, Y, P, Q, S, rsA, rsB, p.lda, p.ldb);
        nt4_ktile<T, false, ABL>(t + 1, nk, L, acc, X, Y, Q, P, S, rsA, rsB, p.lda, p.ldb);
    }
    // mask words of the wave tile (data-gradient launches: 128 rows x 4 words, one 16-byte load per row block and lane),
    // fetched while the last K-tiles multiply - the staging registers are free by then
    const bool masked = p.aux_mode == ASE_AUX_RELU_BITS && bn0 + wc * 128 < p.N;
    i32x4 mrow[4];
    if (masked) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = bm0 + wr * 128 + i * 32 + (lane & 31);
            const int ma = (m >= p.aux_split) ? m - p.aux_delta : m;
            mrow[i] = *reinterpret_cast<const i32x4*>(p.aux + (int64_t)min(ma, p.M - 1) * p.ldaux + ((bn0 + wc * 128) >> 5) * 4);
        }
    }
    for (; t < nk; t += 2) {                         // t is even here: B01(t) sits in P
        nt4_ktile<T, true, ABL>(t, nk, L, acc, X, Y, P, Q, S, rsA, rsB, p.lda, p.ldb);
        if (t + 1 < nk) nt4_ktile<T, true, ABL>(t + 1, nk, L, acc, X, Y, Q, P, S, rsA, rsB, p.lda, p.ldb);
    }
    long long clk1 = 0;
    if constexpr (ABL & 8) clk1 = clock64();
    if (p.prof && tid == 0) p.prof[blockIdx.x * 4 + 2] = wall_clock64();
    uint32_t row_bits[4][4];
    if (masked) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) row_bits[i][j] = (uint32_t)mrow[i][j];
        nt8_epilogue_rows<T, 2, 4>(p, acc, lane, bm0 + wr * 128, bn0 + wc * 128, row_bits);
    } else
        nt8_epilogue_rows<T, 0, 4>(p, acc, lane, bm0 + wr * 128, bn0 + wc * 128, row_bits);
    if (p.prof) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) p.prof[blockIdx.x * 4 + 3] = wall_clock64();
        if constexpr (ABL & 8) {          // "epilogue+drain" column of the lab print x 1000 = shader clocks of the main loop
            if (tid == 0) p.prof[blockIdx.x * 4 + 3] = p.prof[blockIdx.x * 4 + 2] + (unsigned long long)(clk1 - clk0) / 10;
        }
    }
}

}  // namespace

namespace ase_nt {

template <typename T> int launch_nt4r(const NTParams& p0, unsigned long long* prof, hipStream_t stream) {
    constexpr int lds = 2 * 512 * 128;
    static bool attr_done = false;
    auto kern = gemm_nt4r_kernel<T, 0>;
#ifdef ASE_LAB
    static const int abl = getenv("ASE_NT4R_ABL") ? atoi(getenv("ASE_NT4R_ABL")) : 0;
    if (abl == 1) kern = gemm_nt4r_kernel<T, 1>;
    if (abl == 3) kern = gemm_nt4r_kernel<T, 3>;
    if (abl == 4) kern = gemm_nt4r_kernel<T, 4>;
    if (abl == 7) kern = gemm_nt4r_kernel<T, 7>;
    if (abl == 8) kern = gemm_nt4r_kernel<T, 8>;
    if (abl == 15) kern = gemm_nt4r_kernel<T, 15>;
#endif
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) {
            ase_set_error("gemm_nt4r: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
            return ASE_ELAUNCH;
        }
        attr_done = true;
    }
    NTParams p = p0;
    p.prof = prof;
    p.tiles_m = (p.M + 255) / 256;
    p.tiles_n = (p.N + 255) / 256;
    ASE_LAUNCH(kern, dim3(p.tiles_m * p.tiles_n), dim3(256), lds, stream, p);
    ASE_CHECK_LAUNCH("gemm_nt4r");
    return ASE_OK;
}

template int launch_nt4r<bf16_t>(const NTParams&, unsigned long long*, hipStream_t);
template int launch_nt4r<f16_t>(const NTParams&, unsigned long long*, hipStream_t);

}  // namespace ase_nt
