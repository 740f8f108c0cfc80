import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ase_amd.backend import HipBackend
from ase_amd import lib as L
be = HipBackend()
M, N, K = [int(x) for x in sys.argv[1:4]]
dt = torch.bfloat16
A = (torch.randn(M, K, device='cuda') * 0.5).to(dt); B = (torch.randn(N, K, device='cuda') * 0.1).to(dt)
C = torch.zeros(M, N, device='cuda', dtype=dt); bias = torch.randn(N, device='cuda')
for _ in range(10):
    be.gemm_nt(A, B, C, M, N, K, bias=bias, act=L.ACT_RELU)
torch.cuda.synchronize()
