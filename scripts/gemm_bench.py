"""Micro-benchmark of the matrix-core kernels on the layer shapes of the ASE update (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ase_amd.backend import HipBackend
from ase_amd import lib as L
be = HipBackend(x3='--x3' in sys.argv)
dts = ([torch.bfloat16] if '--x3' not in sys.argv else []) + ([torch.float32] if ('--f32' in sys.argv or '--x3' in sys.argv) else [])
NT = [(32768, 1024, 320), (32768, 1024, 1024), (32768, 512, 1024), (32768, 64, 512), (16384, 1024, 1024), (12288, 1024, 1408),
      (12288, 1024, 1024), (12288, 512, 1024), (32768, 512, 64), (32768, 256, 512), (4096, 1408, 1024), (8192, 8192, 8192)]
TN = [(32768, 1024, 1024), (32768, 1024, 320), (32768, 512, 1024), (12288, 1024, 1408), (16384, 1024, 1024), (32768, 64, 512),
      (4096, 1024, 1408)]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for dt in dts:
    print('====', dt)
    for M, N, K in NT:
        A = (torch.randn(M, K, device='cuda') * 0.5).to(dt); B = (torch.randn(N, K, device='cuda') * 0.1).to(dt)
        C = torch.zeros(M, N, device='cuda', dtype=dt); bias = torch.randn(N, device='cuda')
        ms = timeit(lambda: be.gemm_nt(A, B, C, M, N, K, bias=bias, act=L.ACT_RELU))
        print(f'NT {M:6d} x {N:5d} x {K:5d}: {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s')
    for M, N, K in TN:
        A = (torch.randn(M, N, device='cuda') * 0.5).to(dt); B = (torch.randn(M, K, device='cuda') * 0.1).to(dt)
        G = torch.zeros(N, K, device='cuda')
        ms = timeit(lambda: be.gemm_tn(A, B, G, M, N, K, N, K, K, K))
        print(f'TN {M:6d} x {N:5d} x {K:5d}: {ms*1e3:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s')
