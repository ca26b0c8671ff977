"""Experiment (not product): per-segment cycle stamps of the 256x256 ring GEMM, one block, waves 0 and 4."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd._lib import GemmDesc
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ring_probe.so"))
M = N = K = 4096
x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.02).bfloat16()
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
dbg = torch.zeros(8192, device="cuda", dtype=torch.int64)
g = GemmDesc(); g.A, g.B, g.C = x.data_ptr(), w.data_ptr(), out.data_ptr()
g.lda = g.ldb = K; g.ldc = N; g.M, g.N, g.K = M, N, K; g.res_scale = 1.0; g.batch = g.nb0 = 1; g.tune = 302
for mode in (302, 399):
    g.tune = mode
    print('mode', mode, '(399 = stores skipped)')
    for _ in range(3):
        lib.probe_launch(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(g), C.c_void_p(dbg.data_ptr()))
    torch.cuda.synchronize()
    d = dbg.cpu()
    print("wave0: prologue", d[0].item(), "loop", d[1].item(), "epilogue", d[2].item(), "| wave4:", d[4].item(), d[5].item(), d[6].item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.probe_launch(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(g), C.c_void_p(0))
    e1.record(); torch.cuda.synchronize()
    print("kernel us", e0.elapsed_time(e1) / 20 * 1e3)
