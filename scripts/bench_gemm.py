"""Prefill-regime GEMM micro-benchmark (GPU box): the Llama-2-7B projection shapes at M = 4 clips x 702 rows."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2808
tunes = [int(t) for t in sys.argv[2:]] or [0, 300]
shapes = [("qkv", 12288, 4096, 96), ("o", 4096, 4096, 32), ("gu", 22016, 4096, 64), ("down", 4096, 11008, 32), ("clip_fc1", 4096, 1024, 0), ("sq4k", 4096, 4096, 0)]
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K, K2 in shapes:
    m = 4096 if name == "sq4k" else (2056 if name.startswith("clip") else M)
    x = torch.randn(m, K, device="cuda", dtype=BF); w = torch.randn(N, K, device="cuda", dtype=BF) * 0.02
    x2 = torch.randn(m, K2, device="cuda", dtype=BF) if K2 else None
    w2 = torch.randn(N, K2, device="cuda", dtype=BF) if K2 else None
    out = torch.empty(m, N, device="cuda", dtype=BF)
    res = []
    for t in tunes:
        us = timeit(lambda: ops.gemm(x, w, x2=x2, w2=w2, out=out, tune=t))
        res.append(f"tune{t}: {us:7.1f}us {2*m*N*(K+K2)/us/1e6:7.1f} TF/s")
    print(f"{name:9s} M={m} N={N} K={K}+{K2} | " + " | ".join(res), flush=True)
