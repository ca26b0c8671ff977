"""Per-phase s_memtime ticks of the producer / consumer decode panel kernel (experimental build, CRAB_HIP_LIB=scripts/exp/libcrab_timing.so).
consumers (waves 0-7): k step 0 + reads | lgkmcnt wait | barrier | reads of next slot + k step 1;  producers (8-11): vmcnt wait | barrier | issue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
M = 256
for name, N, K, tune in (("gu bn96", 22016, 4096, 79601), ("lm_head bn64", 32017, 4096, 76401)):
    W = torch.randn(N, K, device="cuda", dtype=BF) * 0.02
    x = torch.randn(M, K, device="cuda", dtype=BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    ws = ops._splitk_workspace(x.device)
    for _ in range(3):
        ops.gemm(x, W, out=out, tune=tune)
    torch.cuda.synchronize()
    bn = 96 if tune // 100 % 100 == 96 else 64
    nb = (N + bn - 1) // bn
    t = ws[: nb * 12 * 5 * 8].view(torch.int64).view(nb, 12, 5).double().cpu()
    c = t[:, :8]; pr = t[:, 8:]
    cper = c[..., :4] / c[..., 4:5]
    pper = pr[..., :3] / pr[..., 3:4]
    print(f"{name}: consumers k0 {cper[...,0].mean():.0f} lgkm-wait {cper[...,1].mean():.0f} barrier {cper[...,2].mean():.0f} k1 {cper[...,3].mean():.0f} total {cper.sum(-1).mean():.0f} | "
          f"producers vmcnt-wait {pper[...,0].mean():.0f} barrier {pper[...,1].mean():.0f} issue {pper[...,2].mean():.0f} total {pper.sum(-1).mean():.0f}")
    print("   block 7 consumers:", [[round(v) for v in cper[7, w].tolist()] for w in range(8)])
    print("   block 7 producers:", [[round(v) for v in pper[7, w].tolist()] for w in range(4)])
