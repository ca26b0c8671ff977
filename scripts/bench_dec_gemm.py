"""Decode-regime projections at M = 256 (one row per clip): the batch-tall panel kernel (gemm_decode.hip) per decomposition against
the older split-K kernels, weights rotated through > 600 MB so the 256 MiB MALL cannot hold them.  Times include the reduction kernel.
    python scripts/bench_dec_gemm.py [M] [llama|qwen]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops

BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
which = sys.argv[2] if len(sys.argv) > 2 else "llama"
shapes = {"llama": [("qkv", 12288, 4096, 96, "none"), ("o", 4096, 4096, 32, "none"), ("gu", 22016, 4096, 64, "swiglu_pair"), ("down", 4096, 11008, 32, "none"),
                    ("lm_head", 32017, 4096, 0, "none")],
          "qwen": [("qkv", 4608, 3584, 96, "none"), ("o", 3584, 3584, 32, "none"), ("gu", 37888, 3584, 64, "swiglu_pair"), ("down", 3584, 18944, 32, "none"),
                   ("lm_head", 152081, 3584, 0, "none")]}[which]


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


for name, N, K, K2, act in shapes:
    ncopy = max(2, int(700e6 // (N * K * 2)) + 1)
    Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(ncopy)]
    x = torch.randn(M, K, device="cuda", dtype=BF)
    x2 = torch.randn(M, K2, device="cuda", dtype=BF) if K2 else None
    w2 = torch.randn(N, K2, device="cuda", dtype=BF) * 0.02 if K2 else None
    fp32 = name == "lm_head"
    out = torch.empty(M, N // 2 if act == "swiglu_pair" else N, device="cuda", dtype=torch.float32 if fp32 else BF)
    res = []
    tunes = [("auto", 0), ("old128x4", 104), ("old128x8", 108), ("oldring", 405 if N >= 10240 else 404)]
    for bn in (96, 64):
        for sp in (1, 2, 3, 4, 6, 8):
            tunes.append((f"bn{bn}x{sp}", 70000 + bn * 100 + sp))
    tunes.append(("bn160x1", 91601))
    for label, tune in tunes:
        i = [0]

        def fn():
            i[0] = (i[0] + 1) % ncopy
            ops.gemm(x, Ws[i[0]], x2=x2, w2=w2, out=out, act=act, tune=tune)
        try:
            us = timeit(fn)
        except Exception as e:      # noqa: BLE001
            res.append(f"{label}: ERR")
            continue
        res.append(f"{label}:{us:6.1f}")
    ideal = N * K * 2 / 6.4e6
    print(f"M={M} {name:8s} N={N} K={K}+{K2}  weights@6.4TB/s {ideal:5.1f}us  mfma@1.2PF {2 * M * N * (K + K2) / 1.2e9:5.1f}us | " + "  ".join(res), flush=True)
    del Ws
