"""Two-row-group decode projections (gemm_dec2_kernel, 256 < M <= 512): every (panel width, K slices) decomposition per projection.
    python scripts/bench_dec2_cfg.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops

BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 448
shapes = [("qkv", 12288, 4096, 96, "none"), ("o", 4096, 4096, 32, "none"), ("gu", 22016, 4096, 64, "swiglu_pair"), ("down", 4096, 11008, 32, "none"),
          ("lm_head", 32017, 4096, 0, "none")]


def timeit(fn, n=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, N, K, K2, act in shapes:
    ncopy = max(2, int(700e6 // (N * K * 2)) + 1)
    Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(ncopy)]
    x = torch.randn(M, K, device="cuda", dtype=BF)
    x2 = torch.randn(M, K2, device="cuda", dtype=BF) if K2 else None
    w2 = torch.randn(N, K2, device="cuda", dtype=BF) * 0.02 if K2 else None
    fp32 = name == "lm_head"
    out = torch.empty(M, N // 2 if act == "swiglu_pair" else N, device="cuda", dtype=torch.float32 if fp32 else BF)
    res = []
    for label, tune in [("auto", 0)] + [(f"bn{bn}x{sp}", 70000 + bn * 100 + sp) for bn in (96, 64) for sp in (1, 2, 3, 4, 6)]:
        i = [0]

        def fn():
            i[0] = (i[0] + 1) % ncopy
            ops.gemm(x, Ws[i[0]], x2=x2, w2=w2, out=out, act=act, tune=tune)
        try:
            res.append(f"{label}:{timeit(fn):6.1f}")
        except Exception as e:      # noqa: BLE001
            res.append(f"{label}: ERR")
    print(f"M={M} {name:8s} N={N} K={K}+{K2} | " + "  ".join(res), flush=True)
    del Ws
