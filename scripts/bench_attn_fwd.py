"""Flash-attention forward micro-benchmark (GPU box): decoder prefill shape (16 clips x 32 heads x 702 x 128, causal) and the
CLIP shape (128 frames x 16 heads x 257 x 64).  usage: bench_attn_fwd.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops
BF = torch.bfloat16
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
def run(B, H, S, d, causal):
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(B, S, H * d, device="cuda", generator=g).to(BF)
    k = torch.randn(B, H, S, d, device="cuda", generator=g).to(BF)
    Sp = (S + 7) // 8 * 8
    vt = torch.randn(B, H, d, Sp, device="cuda", generator=g).to(BF)
    o = torch.empty(B, S, H * d, device="cuda", dtype=BF)
    fn = lambda: ops.attn_fwd(q, k, vt, o, q_strides=(S * H * d, d, H * d), k_strides=(H * S * d, S * d, d), vt_strides=(H * d * Sp, d * Sp, Sp),
                              o_strides=(S * H * d, H * d), B=B, H=H, Hk=H, Sq=S, Skv=S, head_dim=d, scale=d ** -0.5, causal=causal)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = 4.0 * B * H * S * S * d * (0.5 if causal else 1.0)
    print(f"attn_fwd B={B} H={H} S={S} d={d} causal={causal}: {us:.1f} us -> {fl/us/1e6:.0f} TFLOP/s", flush=True)
run(16, 32, 702, 128, True)
run(128, 16, 257, 64, False)
