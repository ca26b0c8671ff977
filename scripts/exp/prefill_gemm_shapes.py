"""Every GEMM shape of the prefill phase at its real chunk size and epilogue form (fp32 residual stream where the model has one): us and TFLOP/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def run(tag, M, N, K, act="none", res=None, K2=0):
    x = torch.randn(M, K, device="cuda", dtype=BF); w = torch.randn(N, K, device="cuda", dtype=BF) * 0.02
    b = torch.randn(N, device="cuda", dtype=BF)
    x2 = torch.randn(M, K2, device="cuda", dtype=BF) if K2 else None
    w2 = torch.randn(N, K2, device="cuda", dtype=BF) * 0.02 if K2 else None
    if res == "f32":
        r = torch.randn(M, N, device="cuda", dtype=torch.float32); out = r
    elif res == "bf16":
        r = torch.randn(M, N, device="cuda", dtype=BF); out = r
    else:
        r = None; out = torch.empty(M, N // 2 if act == "swiglu_pair" else N, device="cuda", dtype=BF)
    us = timeit(lambda: ops.gemm(x, w, bias=b if not K2 else None, act=act, residual=r, out=out, x2=x2, w2=w2))
    fl = 2.0 * M * N * (K + K2)
    print(f"{tag:28s} M={M:6d} N={N:5d} K={K:5d}+{K2:2d} res={str(res):5s} {us:8.1f} us {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
Mc = 95 * 8 * 257
for res in ("f32", "bf16"):
    run("clip qkv", Mc, 3072, 1024)
    run("clip out", Mc, 1024, 1024, res=res)
    run("clip fc1", Mc, 4096, 1024, act="quick_gelu")
    run("clip fc2", Mc, 1024, 4096, res=res)
    if res == "bf16": break
Md = 35 * 702
run("dec qkv", Md, 12288, 4096, K2=96)
run("dec o", Md, 4096, 4096, res="f32", K2=32)
run("dec gate|up", Md, 22016, 4096, act="swiglu_pair", K2=64)
run("dec down", Md, 4096, 11008, res="f32", K2=32)
run("dec o (bf16 stream)", Md, 4096, 4096, res="bf16", K2=32)
run("dec down (bf16 stream)", Md, 4096, 11008, res="bf16", K2=32)
run("dec o (no residual)", Md, 4096, 4096, K2=32)
Mb = 95 * 10 * 48
run("beats qkv-ish", Mb, 2304, 768)
run("beats o", Mb, 768, 768, res="f32")
run("beats fc1", Mb, 3072, 768, act="gelu")
run("beats fc2", Mb, 768, 3072, res="f32")
