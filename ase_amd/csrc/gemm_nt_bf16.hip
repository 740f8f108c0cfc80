// NT kernels of the storage type bf16_t (see gemm_nt_kernels.h): the instantiations the library ships.
#include "gemm_nt_kernels.h"

int ase_nt::dispatch_nt_bf16(const NTParams& p, hipStream_t s) { return dispatch_nt<bf16_t>(p, s); }
