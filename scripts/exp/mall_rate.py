"""Experiment (GPU box): LDS-DMA rate per CU from a buffer that fits the 256 MiB MALL but not L2 (every block walks it from its own start)."""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ldsdma_bw.so"))
lib.launch_ldsdma.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
out = torch.zeros(4, device="cuda", dtype=torch.int32)
priv = torch.zeros(1024, device="cuda")
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (2, 16, 64, 128, 192, 512, 2048):
    shared = torch.randn(mb << 20 >> 2, device="cuda")
    for blocks in (256, 64):
        p, depth, iters = 3, 4, 3000
        fn = lambda: lib.launch_ldsdma(torch.cuda.current_stream().cuda_stream, shared.data_ptr(), shared.numel() * 4, priv.data_ptr(), 4096, blocks, iters, 3, p, depth, 0, out.data_ptr())
        ms = t(fn)
        nbytes = blocks * 8 * p * 1024 * iters
        print(f"shared buffer {mb:5d} MiB, blocks={blocks:3d}: {nbytes/ms/1e6/blocks:7.1f} GB/s per CU ({nbytes/ms/1e9:6.2f} TB/s chip)", flush=True)
    del shared
