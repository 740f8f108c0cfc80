#!/bin/bash
# Round 6, GPU call AI: the narrow NT launches (N <= 64: heads, last style layer) as a kernel that FITS BESIDE a phased 256 x 256 workgroup
# (64 x 64 tile, 3-stage ring = 24 KB of LDS, <= 64 registers; libase_hip_c0.so) against the shipped 4-stage / 82-register form: a
# correctness check of the variant, then the same-box A/B of the update (f16gpx3, bf16; three interleaved repetitions).
cd "$(dirname "$0")/../.."
O=gpurun_out/r6ai; mkdir -p $O
python - > $O/check.txt 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
import ase_amd.lib as L
L.LIB_PATH = os.path.join(os.getcwd(), 'ase_amd', 'csrc', 'libase_hip_c0.so')
import torch
from ase_amd.backend import HipBackend
be = HipBackend()
g = torch.Generator().manual_seed(1)
for dt in (torch.float16, torch.bfloat16, torch.float32):
    for M, N, K in ((32768, 64, 512), (4097, 64, 256), (300, 32, 64), (16384, 64, 1024)):
        A = (torch.randn(M, K, generator=g) * 0.5).to(dt).cuda(); B = (torch.randn(N, K, generator=g) * 0.1).to(dt).cuda()
        bias = torch.randn(N, generator=g).cuda()
        C = torch.zeros(M, N, dtype=dt).cuda()
        be.gemm_nt(A, B, C, M, N, K, bias=bias, act=L.ACT_RELU)
        ref = torch.relu(A.float() @ B.float().t() + bias)
        err = float((C.float() - ref).abs().max())
        print(dt, M, N, K, 'kernel', be.lib.ase_hip_gemm_nt_kernel_id(M, N, K, 0 if dt == torch.float32 else 1), 'max err', err)
        assert err <= (2e-2 if dt != torch.float32 else 1e-4) * max(1.0, float(ref.abs().max())), err
print('variant ok')
PY
tail -3 $O/check.txt
REPS=3 timeout 1500 bash scripts/lab/ab_lib.sh libase_hip.so libase_hip_c0.so f16gpx3 > $O/ab_f16gpx3.txt 2>&1; grep update $O/ab_f16gpx3.txt
REPS=3 timeout 1500 bash scripts/lab/ab_lib.sh libase_hip.so libase_hip_c0.so bf16 > $O/ab_bf16.txt 2>&1; grep update $O/ab_bf16.txt
