"""Experiment (GPU box): does touching the first weight bytes of the NEXT small-batch GEMM from the kernel in front of it shorten the GEMM?
Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/exp/prefetch_ramp.hip -o scripts/exp/prefetch_ramp.so"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from crab_amd import ops
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "prefetch_ramp.so"))
lib.launch_touch.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
BF = torch.bfloat16
sink = torch.zeros(4, device="cuda", dtype=torch.int32)
for name, N, K in (("o", 4096, 4096), ("down", 4096, 11008), ("q|k|v", 12288, 4096), ("gate|up", 22016, 4096)):
    nbuf = max(4, int(600e6 / (N * K * 2)) + 1)                     # rotate > 256 MiB of weights: every launch streams from HBM
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(BF) for _ in range(nbuf)]
    x = torch.randn(1, K, device="cuda").to(BF)
    out = torch.empty(1, N, device="cuda", dtype=BF)
    s = torch.cuda.current_stream().cuda_stream
    for kcols in (0, 512, 2048):
        for touch_same in ((False, True) if kcols else (False,)):
            evs = []
            for it in range(3 * nbuf):
                W = Ws[it % nbuf]
                T = W if touch_same else Ws[(it + nbuf // 2) % nbuf]
                if kcols:
                    lib.launch_touch(s, T.data_ptr(), K, N, kcols, sink.data_ptr())
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); ops.gemm(x, W, out=out); e1.record()
                evs.append((e0, e1))
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs[nbuf:])
            print(f"{name:8s} N={N} K={K} touch {kcols:4d} cols of {'THE SAME' if touch_same else 'another '} matrix in front: gemm median {ts[len(ts)//2]:6.1f} us (min {ts[0]:.1f})", flush=True)
