"""Micro-benchmark of the decode-regime kernels over the Llama-2-7B projection shapes (GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops
BF = torch.bfloat16
shapes = [("o", 4096, 4096, 32), ("qkv", 12288, 4096, 96), ("gu", 22016, 4096, 64), ("down", 4096, 11008, 32), ("lm_head", 32017, 4096, 0)]
Ms = [int(a) for a in sys.argv[1:]] or [8, 32, 64, 128]

def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

# rotate among several weight copies so that the 256 MiB infinity cache cannot hold the working set
for M in Ms:
    for name, N, K, K2 in shapes:
        ncopy = max(2, int(600e6 // (N * K * 2)) + 1)
        Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(ncopy)]
        x = torch.randn(M, K, device="cuda", dtype=BF)
        x2 = torch.randn(M, K2, device="cuda", dtype=BF) if K2 else None
        w2 = torch.randn(N, K2, device="cuda", dtype=BF) if K2 else None
        out = torch.empty(M, N, device="cuda", dtype=BF)
        res = []
        for tune in (0, 1, 2, 4, 102, 104, 108, 116, 204, 208, 402, 403, 405, 408, 416):
            if M > 64 and tune == 4: continue
            if M <= 16 and tune >= 100: continue
            if M <= 64 and tune >= 400: continue
            i = [0]
            def fn():
                i[0] = (i[0] + 1) % ncopy
                ops.gemm(x, Ws[i[0]], x2=x2, w2=w2, out=out, tune=tune)
            us = timeit(fn)
            res.append(f"t{tune}:{us:6.1f}us")
        print(f"M={M:4d} {name:8s} ideal@5TB/s {N*K*2/5e6:5.1f}us | " + " ".join(res), flush=True)
        del Ws
    # router
    for K, nproj in ((4096, 3), (4096, 1), (11008, 1)):
        tc = (nproj * 11 + 15) // 16 * 16
        ra = torch.randn(tc, K, device="cuda", dtype=BF) * 0.02
        x = torch.randn(M, K, device="cuda", dtype=BF)
        u = torch.empty(M, 96, device="cuda", dtype=BF)
        wsb = torch.empty(ops.hyperlora_route_workspace(M, K, tc), device="cuda", dtype=torch.uint8)
        us = timeit(lambda: ops.hyperlora_route(x, ra, nproj, 3, 8, (nproj * 24 + 31) // 32 * 32, 2.0, out=u[:, :(nproj * 24 + 31) // 32 * 32], workspace=wsb))
        print(f"M={M:4d} route K={K} nproj={nproj}: {us:6.1f} us (2 launches)", flush=True)
