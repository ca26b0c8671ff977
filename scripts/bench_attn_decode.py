"""Decode-attention micro-benchmark at the bench.py shape (GPU box): B clips x 32 heads x d 128, one layer's KV cache.
usage: bench_attn_decode.py [B] [ctx] [iters] [H] [Hk]  -> us per launch and algorithmic GB/s (K and V rows read once)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = int(sys.argv[2]) if len(sys.argv) > 2 else 830
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
H = int(sys.argv[4]) if len(sys.argv) > 4 else 32
Hk = int(sys.argv[5]) if len(sys.argv) > 5 else H
d = 128; Tmax = 960
g = torch.Generator(device="cuda").manual_seed(1)
kc = (torch.randn(B, Hk, Tmax, d, device="cuda", generator=g) * 0.5).bfloat16()
vc = (torch.randn(B, Hk, Tmax, d, device="cuda", generator=g) * 0.5).bfloat16()
q = torch.randn(B, H * d, device="cuda", generator=g).bfloat16()
o = torch.empty_like(q)
fn = lambda: ops.attn_decode(q, kc, vc, o, B, H, Hk, d, Tmax, ctx, d ** -0.5)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
nbytes = 2.0 * B * ctx * Hk * d * 2 + 2.0 * B * H * d * 2
print(f"attn_decode B={B} ctx={ctx} H={H} Hk={Hk}: {us:.1f} us/launch, algorithmic {nbytes/1e9:.3f} GB -> {nbytes/us/1e3:.0f} GB/s", flush=True)
