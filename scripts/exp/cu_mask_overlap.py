"""Experiment: decode attention (HBM-bound) and a prefill GEMM (MFMA-bound) on CU-masked streams, alone and concurrently.
usage: cu_mask_overlap.py [n_attn_cus]   (the GEMM gets the remaining CUs)"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
hip = C.CDLL("libamdhip64.so")
def masked_stream(bits):
    words = [0] * 8
    for b in bits: words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
NA = int(sys.argv[1]) if len(sys.argv) > 1 else 96
BF = torch.bfloat16
B, ctx, H, d, Tmax = 256, 830, 32, 128, 960
kc = (torch.randn(B, H, Tmax, d, device="cuda") * 0.5).to(BF); vc = (torch.randn(B, H, Tmax, d, device="cuda") * 0.5).to(BF)
q = torch.randn(B, H * d, device="cuda").to(BF); o = torch.empty_like(q)
M, N, K = 11232, 22016, 4096
x = torch.randn(M, K, device="cuda").to(BF); w = (torch.randn(N, K, device="cuda") * 0.02).to(BF); y = torch.empty(M, N, device="cuda", dtype=BF)
def attn(n): 
    for _ in range(n): ops.attn_decode(q, kc, vc, o, B, H, H, d, Tmax, ctx, d ** -0.5)
def gemm(n):
    for _ in range(n): ops.gemm(x, w, out=y)
def timed(fa, fb, sa, sb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_stream(torch.cuda.current_stream()); sb.wait_stream(torch.cuda.current_stream())
    if fa:
        with torch.cuda.stream(sa): fa()
    if fb:
        with torch.cuda.stream(sb): fb()
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
full = torch.cuda.Stream()
attn(2); gemm(2)
NA_IT, NG_IT = 20, 6
timed(lambda: attn(3), lambda: gemm(2), full, full)       # first use of the stream
ta = timed(lambda: attn(NA_IT), None, full, full); tg = timed(None, lambda: gemm(NG_IT), full, full)
print(f"full GPU: attention x{NA_IT} {ta:.2f} ms ({ta/NA_IT*1e3:.0f} us each), gemm x{NG_IT} {tg:.2f} ms ({tg/NG_IT*1e3:.0f} us each); serial sum {ta+tg:.2f} ms")
for layout in ("low", "interleaved"):
    if layout == "low": abits = list(range(NA))
    else: abits = [i for i in range(256) if (i % 8) < (NA * 8 // 256)]          # the same share of every group of 8 bits
    gbits = [i for i in range(256) if i not in set(abits)]
    sa, sb = masked_stream(abits), masked_stream(gbits)
    with torch.cuda.stream(sa): attn(2)
    with torch.cuda.stream(sb): gemm(2)
    torch.cuda.synchronize()
    timed(lambda: attn(3), lambda: gemm(2), sa, sb)
    ta_m = timed(lambda: attn(NA_IT), None, sa, sb); tg_m = timed(None, lambda: gemm(NG_IT), sa, sb)
    tb = timed(lambda: attn(NA_IT), lambda: gemm(NG_IT), sa, sb)
    print(f"mask {layout}: attention on {len(abits)} CUs alone {ta_m:.2f} ms, gemm on {len(gbits)} CUs alone {tg_m:.2f} ms, concurrent {tb:.2f} ms "
          f"(serial full-GPU {ta+tg:.2f} ms)")
