"""Experiment: is the decode attention slower per launch when the step walks the whole 225 GB KV cache (32 layers, one launch each) than when
one layer's 7 GB is re-read?  (translation reach / DRAM page locality of a 225 GB working set; bench: 905-912 us in the step, 876 us in isolation)
usage: attn_decode_footprint.py [clips] [layers]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
B = int(sys.argv[1]) if len(sys.argv) > 1 else 448
L = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx, H, d, Tmax = 830, 32, 128, 960
kv = torch.empty(L, 2, B, H, Tmax, d, device="cuda", dtype=BF)
for l in range(L): kv[l].normal_(0, 0.5)
q = torch.randn(B, H * d, device="cuda").to(BF); o = torch.empty_like(q)
byt = B * H * (ctx * d * 2 * 2) + 2 * B * H * d * 2


def run(layers, n):
    for i in range(n):
        l = layers[i % len(layers)]
        ops.attn_decode(q, kv[l, 0], kv[l, 1], o, B, H, H, d, Tmax, ctx, d ** -0.5)


def timed(layers, n=64):
    run(layers, 8); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(layers, n); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, layers in (("one layer re-read", [0]), ("last layer re-read", [L - 1]), ("two layers", [0, L - 1]), (f"all {L} layers in order", list(range(L)))):
    t = timed(layers)
    print(f"B={B} {name:24s}: {t:7.1f} us per launch, {byt / t / 1e6:6.2f} TB/s", flush=True)
# with the other kernels of a decode layer between the attention launches the caches are cold at every launch: flush with a 512 MB memset
junk = torch.empty(512 << 20, device="cuda", dtype=torch.uint8)
ts = []
for i in range(32):
    junk.zero_(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.attn_decode(q, kv[i % L, 0], kv[i % L, 1], o, B, H, H, d, Tmax, ctx, d ** -0.5); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts.sort()
print(f"single launches after a 512 MB memset + sync: median {ts[len(ts) // 2]:.1f} us, min {ts[0]:.1f}, max {ts[-1]:.1f}")
