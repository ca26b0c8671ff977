"""VERDICT r04 next-6, the cheap form: the decode attention (HBM-bound, matrix pipe idle) and MFMA work (register-only: the ceiling of any GEMM
variant) CO-RESIDENT on the same CUs, on two plain HIP streams.  Measures, at the half-batch shape (224 clips x 32 heads, ctx 830):
  t_attn  = N attention launches alone,  t_mfma = one MFMA-only launch sized to about the same time alone (1 and 2 waves per SIMD),
  t_both  = the two issued together on two streams.   overlap = (t_attn + t_mfma - t_both) / min(t_attn, t_mfma): 1 = free, 0 = serialised.
Kill criterion of the review: concurrent >= 15 % faster than back to back."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "coresident.so"))
lib.launch_mfma_only.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
B, H, d, ctx, Tmax, NL = 224, 32, 128, 830, 960, 20
g = torch.Generator(device="cuda").manual_seed(1)
kc = [(torch.randn(B, H, Tmax, d, device="cuda", generator=g) * 0.5).bfloat16() for _ in range(4)]       # 4 layers' caches (4 x 7 GB x 2): beyond the MALL
vc = [(torch.randn(B, H, Tmax, d, device="cuda", generator=g) * 0.5).bfloat16() for _ in range(4)]
q = torch.randn(B, H * d, device="cuda", generator=g).bfloat16()
o = torch.empty_like(q)
sink = torch.zeros(4, device="cuda")
sa, sm = torch.cuda.Stream(), torch.cuda.Stream()


def attn():
    with torch.cuda.stream(sa):
        for i in range(NL):
            ops.attn_decode(q, kc[i % 4], vc[i % 4], o, B, H, H, d, Tmax, ctx, d ** -0.5)


def mfma(iters, wps):
    lib.launch_mfma_only(sm.cuda_stream, sink.data_ptr(), 256, iters, wps)


def timed(fns, n=5):
    best = 1e9
    for _ in range(n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sa.wait_stream(torch.cuda.current_stream()); sm.wait_stream(torch.cuda.current_stream())
        for f in fns:
            f()
        torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sm)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


attn(); mfma(100, 1); mfma(100, 2); torch.cuda.synchronize()
ta = timed([attn])
nbytes = NL * (2.0 * B * ctx * H * d * 2)
print(f"attention alone: {NL} launches {ta:.2f} ms ({nbytes / ta / 1e6:.0f} GB/s)", flush=True)
for wps in (1, 2):
    # size the MFMA launch to ~ the attention time
    t1 = timed([lambda: mfma(2000, wps)], 3)
    iters = max(100, int(2000 * ta / t1))
    tm = timed([lambda: mfma(iters, wps)])
    fl = 256 * (4 * wps) * iters * 2 * 32 * 16384.0
    tb = timed([attn, lambda: mfma(iters, wps)])
    tb2 = timed([lambda: mfma(iters, wps), attn])
    tb = min(tb, tb2)
    print(f"MFMA-only, {wps} wave(s)/SIMD: alone {tm:.2f} ms ({fl / tm / 1e9:.0f} TFLOP/s) | together {tb:.2f} ms vs back-to-back {ta + tm:.2f} ms "
          f"-> {100 * (1 - tb / (ta + tm)):.1f} % faster, overlap {(ta + tm - tb) / min(ta, tm):.2f}", flush=True)
