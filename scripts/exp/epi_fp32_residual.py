"""o_proj / down_proj at prefill size with the fp32 residual stream in the epilogue (x += proj(att), R == C fp32) against the same GEMM without a
residual: what the R loads cost per 256 x 256 tile.  Run under two builds (CRAB_HIP_LIB) to A/B an epilogue edit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
M = int(sys.argv[1]) if len(sys.argv) > 1 else 24570

def timeit(fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for name, N, K, K2 in (("o", 4096, 4096, 32), ("down", 4096, 11008, 32), ("clip_out", 1024, 1024, 0), ("clip_fc2", 1024, 4096, 0)):
    m = M if not name.startswith("clip") else 131584
    x = torch.randn(m, K, device="cuda", dtype=BF); w = torch.randn(N, K, device="cuda", dtype=BF) * 0.02
    x2 = torch.randn(m, K2, device="cuda", dtype=BF) if K2 else None
    w2 = torch.randn(N, K2, device="cuda", dtype=BF) * 0.02 if K2 else None
    b = torch.randn(N, device="cuda", dtype=BF) if name.startswith("clip") else None
    r32 = torch.randn(m, N, device="cuda"); r16 = r32.to(BF); o16 = torch.empty(m, N, device="cuda", dtype=BF)
    t_plain = timeit(lambda: ops.gemm(x, w, bias=b, x2=x2, w2=w2, out=o16))
    t_bf16 = timeit(lambda: ops.gemm(x, w, bias=b, x2=x2, w2=w2, residual=r16, out=r16))
    t_fp32 = timeit(lambda: ops.gemm(x, w, bias=b, x2=x2, w2=w2, residual=r32, out=r32))
    fl = 2.0 * m * N * (K + K2)
    print(f"{name:8s} M={m} N={N} K={K}+{K2}: no residual {t_plain:7.1f} us ({fl/t_plain/1e6:6.1f} TF/s) | bf16 stream {t_bf16:7.1f} ({fl/t_bf16/1e6:6.1f}) | fp32 stream {t_fp32:7.1f} ({fl/t_fp32/1e6:6.1f})", flush=True)
