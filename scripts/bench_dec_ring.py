"""Decode projections at 256 < M <= 512: the shipped two-row-group panel kernel (auto) against the 256 x 256 ring kernel with SS K slices
(tune 4SS: two row tiles x N / 256 column tiles x SS slices, fp32 slabs + the usual reduction) - is the compute-regime tile the better fit?
    python scripts/bench_dec_ring.py [M]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops

BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 448
shapes = [("qkv", 12288, 4096, 96, "none", (2, 3, 4, 5)), ("o", 4096, 4096, 32, "none", (4, 6, 8)), ("gu", 22016, 4096, 64, "swiglu_pair", (1, 2, 3)),
          ("down", 4096, 11008, 32, "none", (4, 6, 8)), ("lm_head", 32017, 4096, 0, "none", (1, 2))]


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


tot_auto = tot_best = 0.0
for name, N, K, K2, act, splits in shapes:
    ncopy = max(2, int(700e6 // (N * K * 2)) + 1)
    Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(ncopy)]
    x = torch.randn(M, K, device="cuda", dtype=BF)
    x2 = torch.randn(M, K2, device="cuda", dtype=BF) if K2 else None
    w2 = torch.randn(N, K2, device="cuda", dtype=BF) * 0.02 if K2 else None
    fp32 = name == "lm_head"
    out = torch.empty(M, N // 2 if act == "swiglu_pair" else N, device="cuda", dtype=torch.float32 if fp32 else BF)
    res, ts = [], {}
    ref = None
    for label, tune in [("auto", 0)] + [(f"ring x{sp}", 400 + sp) for sp in splits]:
        i = [0]

        def fn():
            i[0] = (i[0] + 1) % ncopy
            ops.gemm(x, Ws[i[0]], x2=x2, w2=w2, out=out, act=act, tune=tune)
        try:
            ts[label] = timeit(fn)
            ops.gemm(x, Ws[0], x2=x2, w2=w2, out=out, act=act, tune=tune)
            if ref is None:
                ref = out.float().clone()
            err = float((out.float() - ref).abs().max() / ref.abs().max())
            res.append(f"{label}:{ts[label]:6.1f} (d {err:.1e})")
        except Exception as e:      # noqa: BLE001
            res.append(f"{label}: ERR {str(e)[:40]}")
    fl = 2.0 * M * N * (K + K2)
    best = min(ts.values())
    k = 1 if name == "lm_head" else 32
    tot_auto += ts["auto"] * k
    tot_best += best * k
    print(f"M={M} {name:8s} N={N} K={K}+{K2} | " + "  ".join(res) + f" | best {fl / best / 1e6:.0f} TFLOP/s", flush=True)
    del Ws
print(f"per decode step (32 layers + lm_head): auto {tot_auto / 1e3:.2f} ms, best-of {tot_best / 1e3:.2f} ms")
