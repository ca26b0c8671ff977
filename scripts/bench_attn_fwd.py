"""Flash-forward micro-benchmark (GPU box): the decoder prefill shape (35 clips x 32 heads x S = 702, d = 128, causal) and the CLIP shape
(280 frames x 16 heads x 257 tokens, d = 64).  CRAB_ATTN_FWD32=0 selects the 64-rows-per-block kernel (run once per setting)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops

BF = torch.bfloat16


def run(name, B, H, Hk, S, hd, causal):
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(B, S, H * hd, device="cuda", generator=g).to(BF)
    k = torch.randn(B, Hk, S, hd, device="cuda", generator=g).to(BF)
    Sp = (S + 7) // 8 * 8
    vt = torch.randn(B, Hk, hd, Sp, device="cuda", generator=g).to(BF)
    o = torch.empty(B, S, H * hd, device="cuda", dtype=BF)

    def go():
        ops.attn_fwd(q, k, vt, o, q_strides=(S * H * hd, hd, H * hd), k_strides=(Hk * S * hd, S * hd, hd), vt_strides=(Hk * hd * Sp, hd * Sp, Sp),
                     o_strides=(S * H * hd, H * hd), B=B, H=H, Hk=Hk, Sq=S, Skv=S, head_dim=hd, scale=hd ** -0.5, causal=causal)
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(20):
        e0.record(); go(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    flops = 4.0 * B * H * S * S * hd * (0.5 if causal else 1.0)
    print(f"{name}: {us:.1f} us (min {ts[0]:.1f}), {flops / us / 1e6:.1f} TFLOP/s algorithmic = {flops / us / 1e6 / 2500 * 100:.1f} % of 2.5 PF", flush=True)


print("CRAB_ATTN_FWD32 =", os.environ.get("CRAB_ATTN_FWD32", "1"))
run("decoder prefill 35 x 32 x 702 d128 causal", 35, 32, 32, 702, 128, True)
run("qwen prefill 35 x 28/4 x 702 d128 causal", 35, 28, 4, 702, 128, True)
run("CLIP 280 x 16 x 257 d64", 280, 16, 16, 257, 64, False)
