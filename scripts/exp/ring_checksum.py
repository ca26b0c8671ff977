"""Experiment helper: bit-level checksums of ring-GEMM outputs over prefill / decode shapes (compare two builds via CRAB_HIP_LIB)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
for (M, N, K, K2) in [(1100, 4096, 4096, 32), (2808, 12288, 4096, 96), (5000, 22016, 4096, 64), (1024, 4096, 11008, 32), (4096, 4096, 4096, 0),
                      (1300, 1024, 1056, 0), (256, 12288, 4096, 96), (256, 22016, 4096, 64), (200, 32017, 4096, 0), (1500, 1024, 1048, 40), (256, 4096, 4096, 32), (256, 4096, 11008, 32), (130, 1000, 1032, 8)]:
    x = torch.randn(M, K, device="cuda", generator=g).to(BF); w = (torch.randn(N, K, device="cuda", generator=g) * 0.02).to(BF)
    x2 = torch.randn(M, K2, device="cuda", generator=g).to(BF) if K2 else None
    w2 = (torch.randn(N, K2, device="cuda", generator=g) * 0.02).to(BF) if K2 else None
    for tune in ((302, 301) if M > 256 else (0, 104, 108)):
        o = ops.gemm(x, w, x2=x2, w2=w2, out_fp32=True, tune=tune)
        print(M, N, K, K2, tune, hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
