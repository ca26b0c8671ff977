"""Experiment: what the fused next-group router costs inside the row-owning split-K reduction (o_proj / down_proj at decode, fp32 residual
stream), and the stand-alone row routers of the o / down groups - the small kernels between the decode projections (4 % of a 448-clip step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 448
N = 4096


def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for K in (4096, 11008):
    x = torch.randn(M, K, device="cuda", dtype=BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    r = torch.randn(M, N, device="cuda"); nw = torch.ones(N, device="cuda", dtype=BF)
    h = torch.empty(M, N, device="cuda", dtype=BF); ra = (torch.randn(48, N, device="cuda") * 0.02).to(BF)
    u = torch.empty(M, 96, device="cuda", dtype=BF)
    res = []
    for name, route in (("no router", None), ("2 projections (gate|up)", (ra, 2, 3, 8, 64, 2.0, u)), ("3 projections (q|k|v)", (ra, 3, 3, 8, 96, 2.0, u))):
        t = timeit(lambda: ops.gemm(x, w, residual=r, out=r, post_norm=(nw, 1e-5, h), route=route))
        res.append(f"{name}: {t:.1f}")
    print(f"M={M} K={K} GEMM + row-owning reduction + norm, us: " + " | ".join(res), flush=True)
for K, nm in ((4096, "o group (x = attention output)"), (11008, "down group (x = SwiGLU output)")):
    x = torch.randn(M, K, device="cuda", dtype=BF); ra = (torch.randn(16, K, device="cuda") * 0.02).to(BF)
    uo = torch.empty(M, 32, device="cuda", dtype=BF); wsb = torch.empty(ops.hyperlora_route_workspace(M, K, 16) + 256, device="cuda", dtype=torch.uint8)
    t = timeit(lambda: ops.hyperlora_route(x, ra, 1, 3, 8, 32, 2.0, out=uo, workspace=wsb))
    print(f"M={M} stand-alone row router, {nm}: {t:.1f} us", flush=True)
