"""Experiment: latency of one ring-kernel block on short K (what a split-K slice of a decode GEMM would cost)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K in (("qkv/5", 12288, 832), ("gu/2", 22016, 2080), ("gu/1", 22016, 4160), ("down/16", 4096, 704), ("o/16", 4096, 256), ("o/8", 4096, 512)):
    # rotate weights so that they come from HBM like in the decode loop
    ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
    Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(ncopy)]
    x = torch.randn(512, K, device="cuda", dtype=BF)      # M = 512: two full row tiles (M <= 256 would take the decode dispatch)
    out = torch.empty(512, N, device="cuda", dtype=torch.float32)
    i = [0]
    def fn():
        i[0] = (i[0] + 1) % ncopy
        ops.gemm(x, Ws[i[0]], out=out, tune=302)
    print(f"{name:8s} N={N} K={K}: {timeit(fn):6.1f} us ({2 * ((N + 255) // 256)} blocks, fp32 out)", flush=True)
