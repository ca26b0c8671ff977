"""r06: where should the decode projections switch from the 128 x 128 split-K kernel to the batch-tall panel kernel?  The four projections of a
Llama-2-7B layer + lm_head at M rows (weights rotated through > 600 MB), automatic kernel choice, under CRAB_DEC_MIN_ROWS = 128 (r02-r05: the
panel kernel only above 128 rows) and = 64.  Run once per setting: CRAB_DEC_MIN_ROWS=64 python scripts/exp/dec_min_rows.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops

BF = torch.bfloat16
shapes = [("qkv", 12288, 4096, 96, "none"), ("o", 4096, 4096, 32, "none"), ("gu", 22016, 4096, 64, "swiglu_pair"), ("down", 4096, 11008, 32, "none"), ("lm_head", 32017, 4096, 0, "none")]


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


print("CRAB_DEC_MIN_ROWS =", os.environ.get("CRAB_DEC_MIN_ROWS", "128 (default)"))
for M in ([int(x) for x in sys.argv[1:]] or [72, 80, 96, 112, 128, 160, 256]):
    tot, row = 0.0, []
    for name, N, K, K2, act in shapes:
        ncopy = max(2, int(700e6 // (N * K * 2)) + 1)
        Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(ncopy)]
        x = torch.randn(M, K, device="cuda", dtype=BF)
        x2 = torch.randn(M, K2, device="cuda", dtype=BF) if K2 else None
        w2 = torch.randn(N, K2, device="cuda", dtype=BF) * 0.02 if K2 else None
        out = torch.empty(M, N // 2 if act == "swiglu_pair" else N, device="cuda", dtype=torch.float32 if name == "lm_head" else BF)
        i = [0]

        def fn():
            i[0] = (i[0] + 1) % ncopy
            ops.gemm(x, Ws[i[0]], x2=x2, w2=w2, out=out, act=act)
        us = timeit(fn)
        row.append(f"{name} {us:6.1f}")
        tot += us * (1 if name == "lm_head" else 32)
        del Ws
    print(f"M={M:4d}  " + "  ".join(row) + f"   | 32 layers + lm_head: {tot / 1e3:6.2f} ms", flush=True)
