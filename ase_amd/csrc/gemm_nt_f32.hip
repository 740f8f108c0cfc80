// NT kernels of the storage type float (see gemm_nt_kernels.h): the instantiations the library ships.
#include "gemm_nt_kernels.h"

int ase_nt::dispatch_nt_f32(const NTParams& p, hipStream_t s) { return dispatch_nt<float>(p, s); }
