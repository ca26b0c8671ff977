"""The four decoder projections at M = 256 through the automatic decode path (gemm_dec_kernel + the fused reductions that follow it in
the model are replaced here by the plain reduction): a few launches each, for rocprofv3 --pmc passes (scripts/pmc_fetch.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops

BF = torch.bfloat16
M = 256
for name, N, K, K2, act in (("qkv", 12288, 4096, 96, "none"), ("o", 4096, 4096, 32, "none"), ("gu", 22016, 4096, 64, "swiglu_pair"), ("down", 4096, 11008, 32, "none")):
    Ws = [torch.randn(N, K, device="cuda", dtype=BF) * 0.02 for _ in range(8)]
    x, x2, w2 = torch.randn(M, K, device="cuda", dtype=BF), torch.randn(M, K2, device="cuda", dtype=BF), torch.randn(N, K2, device="cuda", dtype=BF) * 0.02
    out = torch.empty(M, N // 2 if act == "swiglu_pair" else N, device="cuda", dtype=BF)
    for i in range(8):
        ops.gemm(x, Ws[i], x2=x2, w2=w2, out=out, act=act)
    torch.cuda.synchronize()
    print(name, "weights MiB", N * (K + K2) * 2 / 2**20)
