"""Decode projections over TWO 256-row groups in one launch (gemm_decode.hip, GemmDP.groups) against two separate 256-row launches:
does a weight panel cost one HBM pass for 512 rows?  Weights rotated through > 600 MB (the 256 MiB MALL cannot hold them).
    python scripts/bench_dec_gemm_groups.py            # CRAB_DEC_GROUPS_NT=1: non-temporal weight loads in the two-group form"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops

BF = torch.bfloat16
shapes = [("qkv", 12288, 4096, 96, "none"), ("o", 4096, 4096, 32, "none"), ("gu", 22016, 4096, 64, "swiglu_pair"), ("down", 4096, 11008, 32, "none"),
          ("lm_head", 32017, 4096, 0, "none")]


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {}
for name, N, K, K2, act in shapes:
    ncopy = max(2, int(700e6 // (N * K * 2)) + 1)
    Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(ncopy)]
    w2 = torch.randn(N, K2, device="cuda", dtype=BF) * 0.02 if K2 else None
    fp32 = name == "lm_head"
    row = []
    for M in (256, 320, 384, 448, 512):
        x = torch.randn(M, K, device="cuda", dtype=BF)
        x2 = torch.randn(M, K2, device="cuda", dtype=BF) if K2 else None
        out = torch.empty(M, N // 2 if act == "swiglu_pair" else N, device="cuda", dtype=torch.float32 if fp32 else BF)
        i = [0]

        def one():
            i[0] = (i[0] + 1) % ncopy
            ops.gemm(x, Ws[i[0]], x2=x2, w2=w2, out=out, act=act)

        def two():                      # the same rows as two launches of <= 256 rows (what r03 would do for a second decode group)
            i[0] = (i[0] + 1) % ncopy
            ops.gemm(x[:256], Ws[i[0]], x2=x2[:256] if K2 else None, w2=w2, out=out[:256], act=act)
            if M > 256:
                ops.gemm(x[256:], Ws[i[0]], x2=x2[256:] if K2 else None, w2=w2, out=out[256:], act=act)
        a, b = timeit(one), timeit(two)
        tot.setdefault(M, [0.0, 0.0])
        k = 1 if name == "lm_head" else 32
        tot[M][0] += a * k; tot[M][1] += b * k
        row.append(f"M={M}: one {a:6.1f} us  split {b:6.1f} us  ({a / M * 256:5.1f} us per 256 rows)")
    print(f"{name:8s} N={N} K={K}+{K2} | " + " | ".join(row), flush=True)
    del Ws
for M, (a, b) in tot.items():
    print(f"per decode step (32 layers + lm_head) M={M}: one launch {a / 1e3:6.2f} ms, split launches {b / 1e3:6.2f} ms, per 256 rows {a / 1e3 / M * 256:6.2f} ms")
