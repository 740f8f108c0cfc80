// LAB build of the bf16 NT dispatch (NOT part of the product: scripts/lab/Makefile links it into libase_hip_lab.so in place of
// ase_amd/csrc/gemm_nt_bf16.o).  Same kernels as the product (the templates of ase_amd/csrc/gemm_nt_kernels.h); what the lab adds
// is a dispatch that tuning runs steer through environment variables, read once:
//   ASE_NT_VARIANT=n   force one of the co-resident / skinny tilings below for every bf16 launch
//   ASE_NT_TILE=128|256, ASE_NT_PHASED=0   force the tile class / keep the phased kernel out
//   ASE_NT4R=1         the 4-wave register-staged 256 x 256 kernel (gemm_nt4r_variant.hip) where its epilogue applies
//   ASE_NT4V=<bits>    the 4-wave LDS-DMA kernel with the half-K-tile pipeline (gemm_nt4v_variant.hip; bits = its ablation switches)
// The schedule ablations of the phased kernel (no DMA / no reads / no MFMAs / 16x16x32 MFMA shape / DMA in the read half: timing
// only, wrong results) lived behind `#ifdef ASE_LAB` in the product's gemm.hip up to commit 9f99095 and were removed with it;
// their results are kept in profiles/r03_lab_*.log and profiles/r04_lab_mfma_shape.txt.
#include "../../ase_amd/csrc/gemm_nt_kernels.h"

namespace ase_nt {
template <typename T> int launch_nt4r(const NTParams& p, unsigned long long* prof, hipStream_t stream);      // gemm_nt4r_variant.hip
template <typename T> int launch_nt4v(const NTParams& p, unsigned long long* prof, int variant, hipStream_t stream);   // gemm_nt4v_variant.hip
}

static int lab_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

int ase_nt::dispatch_nt_bf16(const NTParams& p, hipStream_t s) {
    typedef bf16_t T;
    const bool k128 = (p.K * (int)sizeof(T)) % 128 == 0;
    static const int variant = lab_knob("ASE_NT_VARIANT", 0), force = lab_knob("ASE_NT_TILE", 0), phased = lab_knob("ASE_NT_PHASED", 1),
                     nt4r = lab_knob("ASE_NT4R", 0), nt4v = lab_knob("ASE_NT4V", -1);
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
    int choice = nt_choice(p.M, p.N, p.K, 2, true);
    if (force == 256 && p.N > 64) choice = k128 ? 2 : 3;
    if (force == 128 && p.N > 64) choice = 1;
    if (choice == 2 && !phased) choice = 3;
    if ((choice == 2 || choice == 6) && nt4v >= 0 && rows_epi(p, 128) && p.pre_out == nullptr) return launch_nt4v<T>(p, g_nt_prof, nt4v, s);
    if (choice == 2 && nt4r && rows_epi(p, 128) && p.pre_out == nullptr) return launch_nt4r<T>(p, g_nt_prof, s);
    if (choice == 2) return rows_epi(p, 64) ? launch_nt8<T, true>(p, s) : launch_nt8<T, false>(p, s);
    if (choice == 3) return k128 ? launch_nt<T, 4, 2, 2, 4, 128, 2>(p, s) : launch_nt<T, 4, 2, 2, 4, 64, 4>(p, s);
    return dispatch_nt<T>(p, s);          // the product's choice for everything else
}
