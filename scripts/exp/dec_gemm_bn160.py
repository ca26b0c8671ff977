import sys, os
sys.path.insert(0, "/root/repo")
import torch
from crab_amd import ops
BF = torch.bfloat16
torch.manual_seed(0)
M, N, K, K2 = 256, 37888, 3584, 64
x = torch.randn(M, K, device="cuda", dtype=BF); W = torch.randn(N, K, device="cuda", dtype=BF) * 0.02
x2 = torch.randn(M, K2, device="cuda", dtype=BF); w2 = torch.randn(N, K2, device="cuda", dtype=BF) * 0.02
for act in ("swiglu_pair", "none"):
    outs = {}
    for tune in (79601, 91601, 0):
        out = torch.empty(M, N // 2 if act == "swiglu_pair" else N, device="cuda", dtype=BF)
        ops.gemm(x, W, x2=x2, w2=w2, out=out, act=act, tune=tune)
        outs[tune] = out.float()
    ref = x.float() @ W.float().t() + x2.float() @ w2.float().t()
    if act == "swiglu_pair":
        ref = torch.nn.functional.silu(ref[:, 0::2]) * ref[:, 1::2]
    for t, o in outs.items():
        print(act, t, "max err vs fp32", (o - ref).abs().max().item(), "equal to 96-panel:", torch.equal(o, outs[79601]))
import time
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(4)]
out = torch.empty(M, N // 2, device="cuda", dtype=BF)
for tune in (79601, 91601, 0, 79601, 91601, 0):
    i = [0]
    def fn():
        i[0] = (i[0] + 1) % 4
        ops.gemm(x, Ws[i[0]], x2=x2, w2=w2, out=out, act="swiglu_pair", tune=tune)
    print("tune", tune, f"{timeit(fn):.1f} us")
