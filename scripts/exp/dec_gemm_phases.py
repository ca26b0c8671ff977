"""Per-phase cycle counts of the decode panel kernel's steady-state loop (experimental build with s_memtime stamps: CRAB_HIP_LIB=
scripts/exp/libcrab_timing.so).  Phases per 64-wide slot and wave: k step 0 (12 MFMAs + the reads of k step 1), counted vmcnt wait,
barrier, k step 1 (reads of the next slot, 12 MFMAs interleaved with the refill LDS-DMA)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
M = 256
for name, N, K, tune in (("gu", 22016, 4096, 79601), ("lm_head/bn64", 32017, 4096, 76401)):
    W = torch.randn(N, K, device="cuda", dtype=BF) * 0.02
    x = torch.randn(M, K, device="cuda", dtype=BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    ws = ops._splitk_workspace(x.device)
    for _ in range(3):
        ops.gemm(x, W, out=out, tune=tune)
    torch.cuda.synchronize()
    bn = 96 if tune // 100 % 100 == 96 else 64
    nb = (N + bn - 1) // bn
    t = ws[: nb * 8 * 5 * 8].view(torch.int64).view(nb, 8, 5).double().cpu()
    it = t[..., 4]
    per = t[..., :4] / it[..., None]
    print(f"{name}: blocks {nb}, steady iterations {it[0,0].item():.0f}; memtime ticks per slot (mean over blocks / waves): "
          f"kstep0 {per[...,0].mean():.0f}  vmcnt-wait {per[...,1].mean():.0f}  barrier {per[...,2].mean():.0f}  kstep1+refill {per[...,3].mean():.0f}  total {per.sum(-1).mean():.0f}")
    print("   per wave (block 0):", [[round(v) for v in per[0, w].tolist()] for w in range(8)])
    print("   per wave (block 100):", [[round(v) for v in per[100, w].tolist()] for w in range(8)])
