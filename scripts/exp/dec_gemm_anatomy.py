import os, sys
sys.path.insert(0, '/root/repo')
import torch
from crab_amd import ops
BF = torch.bfloat16
M = 256
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, N, K, tunes in (("o", 4096, 4096, (76401, 79601)), ("gu", 22016, 4096, (79601, 76401)), ("qkv", 12288, 4096, (79601, 79602))):
    ncopy = max(2, int(700e6 // (N * K * 2)) + 1)
    Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(ncopy)]
    x = torch.randn(M, K, device="cuda", dtype=BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    for tune in tunes:
        i = [0]
        def fn():
            i[0] = (i[0] + 1) % ncopy
            ops.gemm(x, Ws[i[0]], out=out, tune=tune)
        print(f"DBG={os.environ.get('CRAB_DEC_DBG','0'):3s} {name:4s} tune={tune}: {timeit(fn):7.1f} us", flush=True)
    del Ws
