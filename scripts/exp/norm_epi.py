"""Experiment: cost split of the row-owning split-K reduction (o_proj / down_proj at decode): with and without the fused router."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
M, N = 256, 4096
for K in (4096, 11008):
    x = torch.randn(M, K, device="cuda", dtype=BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    r = torch.randn(M, N, device="cuda", dtype=BF); nw = torch.ones(N, device="cuda", dtype=BF)
    h = torch.empty(M, N, device="cuda", dtype=BF); ra = (torch.randn(48, N, device="cuda") * 0.02).to(BF)
    u = torch.empty(M, 96, device="cuda", dtype=BF)
    for route in (None, (ra, 3, 3, 8, 96, 2.0, u)):
        for _ in range(20):
            ops.gemm(x, w, residual=r, out=r, post_norm=(nw, 1e-5, h), route=route)
    torch.cuda.synchronize()
