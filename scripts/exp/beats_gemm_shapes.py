"""BEATs projection shapes at 256 clips (M = 256 x 10 windows x 48 tokens = 122880 rows, width 768): the 128x128 two-stage kernel (tune 301)
against the 256x256 ring kernel (tune 302), whose automatic rule asks for K >= 1024."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
M = 122880
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K, act in (("qkv", 2304, 768, "none"), ("o", 768, 768, "none"), ("fc1", 3072, 768, "gelu"), ("fc2", 768, 3072, "none")):
    x = torch.randn(M, K, device="cuda", dtype=BF); w = torch.randn(N, K, device="cuda", dtype=BF) * 0.02
    b = torch.randn(N, device="cuda", dtype=BF); out = torch.empty(M, N, device="cuda", dtype=BF)
    res = []
    for t in (0, 301, 302):
        us = timeit(lambda: ops.gemm(x, w, bias=b, act=act, out=out, tune=t))
        res.append(f"tune{t}: {us:7.1f} us {2 * M * N * K / us / 1e6:7.1f} TF/s")
    print(f"{name:4s} N={N} K={K} | " + " | ".join(res), flush=True)
