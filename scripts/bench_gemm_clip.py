"""CLIP ViT-L/14 projection shapes at M = 16 clips x 8 frames x 257 tokens (GPU box): ring (302) vs two-stage 128^2 (301)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32896
tunes = [int(t) for t in sys.argv[2:]] or [302, 301]
shapes = [("qkv", 3072, 1024, None), ("o", 1024, 1024, None), ("fc1", 4096, 1024, "quick_gelu"), ("fc2", 1024, 4096, None)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K, act in shapes:
    x = torch.randn(M, K, device="cuda", dtype=BF); w = torch.randn(N, K, device="cuda", dtype=BF) * 0.02
    b = torch.randn(N, device="cuda", dtype=BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    res = []
    for t in tunes:
        us = timeit(lambda: ops.gemm(x, w, bias=b, act=act, out=out, tune=t))
        res.append(f"tune{t}: {us:7.1f}us {2*M*N*K/us/1e6:7.1f} TF/s")
    print(f"{name:5s} M={M} N={N} K={K} act={act} | " + " | ".join(res), flush=True)
