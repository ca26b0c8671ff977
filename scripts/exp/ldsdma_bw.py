"""Experiment (GPU box): per-CU throughput of the LDS-DMA path from L2-resident data, from HBM, and mixed 2:1 (scripts/exp/ldsdma_bw.hip).
Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/exp/ldsdma_bw.hip -o scripts/exp/ldsdma_bw.so"""
import ctypes as C, os, torch
LIBPATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ldsdma_bw.so")
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ldsdma_bw.so"))
lib.launch_ldsdma.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
shared = torch.randn(2 << 20 >> 2, device="cuda")                       # 2 MiB, L2 resident
ppb = 16 << 20
priv = torch.randn(256 * ppb >> 2, device="cuda")                       # 4 GiB: 16 MiB per block
out = torch.zeros(4, device="cuda", dtype=torch.int32)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for blocks in (256, 64):
    for mode, name in ((0, "L2-resident shared 2 MiB"), (1, "HBM private stream"), (2, "2 shared : 1 private")):
        for p, depth in ((3, 2), (3, 4), (3, 6), (6, 2), (6, 3)):
            for nt in ((0, 1) if mode else (0,)):
                iters = 2000 if mode != 1 else (ppb // (8 * p * 1024)) - 2
                fn = lambda: lib.launch_ldsdma(torch.cuda.current_stream().cuda_stream, shared.data_ptr(), shared.numel() * 4, priv.data_ptr(), ppb, blocks, iters, mode, p, depth, nt, out.data_ptr())
                ms = t(fn)
                nbytes = blocks * 8 * p * 1024 * iters
                print(f"blocks={blocks:3d} {name:26s} pieces/wave/batch={p} depth={depth} nt={nt}: {nbytes/ms/1e6/blocks:7.1f} GB/s per CU  ({nbytes/ms/1e9:6.2f} TB/s chip)", flush=True)
import ctypes as C, os, torch

lib.launch_vload.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
K = 4096
buf = torch.randn(256 * K // 2, device="cuda")      # [256][4096] bf16 = 2 MiB
out = torch.zeros(512, device="cuda", dtype=torch.int32)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for blocks in (256, 64):
    for frag in (1, 0):
        for un in (1, 2, 4):
            iters = 4000 // un
            fn = lambda: lib.launch_vload(torch.cuda.current_stream().cuda_stream, buf.data_ptr(), K, blocks, iters, frag, un, out.data_ptr())
            ms = t(fn)
            nbytes = blocks * 8 * 4 * 1024 * un * iters
            print(f"vector loads blocks={blocks:3d} {'fragment-shaped (16 rows x 64 B)' if frag else 'whole lines (8 rows x 128 B)   '} slots in flight={un}: {nbytes/ms/1e6/blocks:7.1f} GB/s per CU ({nbytes/ms/1e9:6.2f} TB/s chip)", flush=True)
