"""Experiment (GPU box): LDS-DMA from L2 on four waves + ordinary non-temporal loads from HBM on four other waves of the same block: do the two
paths share one per-CU in-flight limit (scripts/exp/dual_path.hip)?  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC dual_path.hip -o dual_path.so"""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "dual_path.so"))
lib.launch_dual.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
shared = torch.randn(2 << 20 >> 2, device="cuda")                       # 2 MiB, L2 resident
ppb = 16 << 20
priv = torch.randn(256 * ppb >> 2, device="cuda")                       # 4 GiB: 16 MiB per block
out = torch.zeros(512, device="cuda", dtype=torch.int32)
def t(fn, n=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
CFG = {0: (8, 3, 2), 1: (8, 3, 4), 2: (4, 3, 4), 3: (8, 6, 4)}
for blocks in (256,):
    for cfg, (pa, pb, depth) in CFG.items():
        iters = 1200
        res = {}
        for which, name in ((1, "DMA(L2) alone"), (2, "loads(HBM) alone"), (3, "both")):
            fn = lambda: lib.launch_dual(torch.cuda.current_stream().cuda_stream, shared.data_ptr(), shared.numel() * 4, priv.data_ptr(), ppb, blocks, iters, which, cfg, out.data_ptr())
            ms = t(fn)
            a = blocks * 4 * pa * 1024 * iters if which & 1 else 0
            b = blocks * 4 * pb * 1024 * iters if which & 2 else 0
            res[which] = ms
            print(f"blocks={blocks} PA={pa} PB={pb} depth={depth} {name:18s}: {ms*1e3/iters:6.3f} us per batch | DMA {a/ms/1e6/blocks:6.1f} GB/s per CU, loads {b/ms/1e6/blocks:6.1f} GB/s per CU ({b/ms/1e9:5.2f} TB/s chip)", flush=True)
