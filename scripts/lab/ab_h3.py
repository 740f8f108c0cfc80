"""Time the six launches of the gradient penalty's value path (ASE_F32H3: f32 A split in registers, pre-split packed B) for one build
of the library: python scripts/lab/ab_h3.py <libname in ase_amd/csrc>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ase_amd.lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1])
import torch
from ase_amd.backend import HipBackend
be = HipBackend(x3='f16')
def timeit(fn, n=30):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
out, tot = [], 0.0
for M, N, K, ea in [(4096, 1024, 1408, 12), (4096, 1024, 1024, 6), (4096, 512, 1024, 6), (4096, 1024, 512, 12), (4096, 1024, 1024, 12), (4096, 1408, 1024, 12)]:
    A = torch.randn(M, K, device='cuda') * (1.0 if ea == 12 else 0.5)
    W = torch.randn(N, K, device='cuda') * 0.03
    Bs = torch.zeros((N + 63) // 64 * 64, K, device='cuda')
    be.refresh_shadow(W, Bs, None, K, K, x3_exp=11)
    C = torch.zeros(M, (N + 63) // 64 * 64, device='cuda'); bias = torch.randn(N, device='cuda')
    bits = torch.zeros(M, C.shape[1] // 32, dtype=torch.int32, device='cuda')
    ms = timeit(lambda: be.gemm_nt(A, Bs, C, M, C.shape[1], K, bias=None, act=L.ACT_RELU, mask_out=bits, x3_exps=(ea, 11)))
    ref = torch.relu(A.double() @ W.double().t())
    err = float((C[:, :N].double() - ref).abs().max() / ref.abs().max())
    out.append(f'{M}x{N}x{K} {ms * 1e3:.1f}us (err {err:.1e})'); tot += ms * 1e3
print(sys.argv[1], ' | '.join(out), f'| sum {tot:.1f} us', flush=True)
