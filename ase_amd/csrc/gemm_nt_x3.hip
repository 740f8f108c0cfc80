// NT kernels of the storage type f32s_t (see gemm_nt_kernels.h): the instantiations the library ships.
#include "gemm_nt_kernels.h"

int ase_nt::dispatch_nt_x3(const NTParams& p, hipStream_t s) { return dispatch_nt<f32s_t>(p, s); }
