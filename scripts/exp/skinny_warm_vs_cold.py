"""Does a decode projection at M <= 16 run faster when its weights (or the head of them) are already cache-resident?  (input to the
"prefetch the next projection's weights during the layer tail" idea)  Same GEMM, weights rotated through 700 MB (cold: HBM)
vs ONE copy re-used (warm: 32-180 MB, inside the 256 MB MALL, partly in the 32 MB of L2)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
shapes = [("o", 4096, 4096), ("qkv", 12288, 4096), ("gu", 22016, 4096), ("down", 4096, 11008)]

def timeit(fn, n=60):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M in (1, 8):
    for name, N, K in shapes:
        ncopy = int(700e6 // (N * K * 2)) + 2
        Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(ncopy)]
        x = torch.randn(M, K, device="cuda", dtype=BF)
        out = torch.empty(M, N, device="cuda", dtype=BF)
        i = [0]
        def cold():
            i[0] = (i[0] + 1) % ncopy
            ops.gemm(x, Ws[i[0]], out=out)
        def warm():
            ops.gemm(x, Ws[0], out=out)
        c, w = timeit(cold), timeit(warm)
        print(f"M={M} {name:5s} {N*K*2/1e6:6.1f} MB: cold {c:6.1f} us ({N*K*2/c/1e6:5.2f} TB/s)   warm {w:6.1f} us ({N*K*2/w/1e6:5.2f} TB/s)", flush=True)
        del Ws
