"""Experiment (GPU box): LDS-DMA per-CU throughput by piece shape from an L2-resident [256][4096] bf16 matrix: 16 rows x 64 B (half lines,
32-wide K tiles) vs 8 rows x 128 B (whole lines, 64-wide K slots).  scripts/exp/ldsdma_bw.hip"""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ldsdma_bw.so"))
lib.launch_ldsdma_tile.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
K = 4096
buf = torch.randn(256 * K // 2, device="cuda")
out = torch.zeros(4, device="cuda", dtype=torch.int32)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for blocks in (256, 64):
    for half in (1, 0):
        for depth in (2, 4):
            iters = 2000
            ms = t(lambda: lib.launch_ldsdma_tile(torch.cuda.current_stream().cuda_stream, buf.data_ptr(), K, blocks, iters, half, depth, out.data_ptr()))
            nbytes = blocks * 8 * 4 * 1024 * iters
            print(f"blocks={blocks:3d} pieces of {'16 rows x 64 B ' if half else ' 8 rows x 128 B'} depth={depth}: {nbytes/ms/1e6/blocks:7.1f} GB/s per CU ({nbytes/ms/1e9:6.2f} TB/s chip)", flush=True)
