// NT kernels of the storage type f32h_t (see gemm_nt_kernels.h): the instantiations the library ships.
#include "gemm_nt_kernels.h"

int ase_nt::dispatch_nt_h3(const NTParams& p, hipStream_t s) { return dispatch_nt<f32h_t>(p, s); }
