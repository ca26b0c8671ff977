"""gate|up projection at prefill size: plain epilogue + swiglu kernel vs fused swiglu-pair epilogue (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 11232
N, K, K2 = 22016, 4096, 64
x = torch.randn(M, K, device="cuda", dtype=BF); w = torch.randn(N, K, device="cuda", dtype=BF) * 0.02
x2 = torch.randn(M, K2, device="cuda", dtype=BF); w2 = torch.randn(N, K2, device="cuda", dtype=BF) * 0.02
gu = torch.empty(M, N, device="cuda", dtype=BF); act = torch.empty(M, N // 2, device="cuda", dtype=BF)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for tune in (302, 301):
    a = timeit(lambda: ops.gemm(x, w, x2=x2, w2=w2, out=gu, tune=tune))
    b = timeit(lambda: ops.swiglu(gu, out=act))
    c = timeit(lambda: ops.gemm(x, w, x2=x2, w2=w2, out=act, act="swiglu_pair", tune=tune))
    print(f"M={M} tune{tune}: gemm {a:.1f} us + swiglu {b:.1f} us = {a+b:.1f} | fused {c:.1f} us", flush=True)
