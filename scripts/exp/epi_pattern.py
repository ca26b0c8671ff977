"""r06 experiment: the fp32-residual epilogue of the ring GEMM's 256x256 tile by address pattern alone (scripts/exp/epi_pattern.hip; build:
hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/exp/epi_pattern.hip -o scripts/exp/epi_pattern.so)."""
import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "epi_pattern.so"))
lib.launch_epi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]


def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, what in ((24576, 4096, "decoder o / down"), (195584, 1024, "CLIP out / fc2")):
    R = torch.randn(M, N, device="cuda")
    Cc = torch.empty_like(R)
    for gap in (0,):
        for pat in (0, 1):
            us = t(lambda: lib.launch_epi(torch.cuda.current_stream().cuda_stream, R.data_ptr(), Cc.data_ptr(), M, N, pat, gap, 256))
            rounds = (M // 256) * (N // 256) / 256
            print(f"{what:18s} [{M} x {N}] gap={gap:3d} pattern {pat}: {us:8.1f} us = {us / rounds:6.2f} us per round of 256 tiles, {2 * M * N * 4 / us / 1e6:6.2f} TB/s")

# how much of the per-round cost is the chip's copy rate (falls with fewer CUs bursting at once) and how much is per CU (does not)?
M, N = 24576, 4096
R = torch.randn(M, N, device="cuda"); Cc = torch.empty_like(R)
for blocks in (256, 192, 128, 64, 32, 8):
    for pat in (0, 1):
        us = t(lambda: lib.launch_epi(torch.cuda.current_stream().cuda_stream, R.data_ptr(), Cc.data_ptr(), M, N, pat, 0, blocks))
        per_tile = us / ((M // 256) * (N // 256) / blocks)
        print(f"[{M} x {N}] {blocks:3d} blocks, pattern {pat}: {per_tile:6.2f} us per tile and block ({512 * 1024 / per_tile / 1e3:6.1f} GB/s per CU), chip {2 * M * N * 4 / us / 1e6:5.2f} TB/s")
