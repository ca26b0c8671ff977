"""Grid-barrier cost in a persistent kernel (one block per CU) against kernel boundaries (GPU box).  Build:
hipcc --offload-arch=gfx950 -O3 -shared -fPIC grid_barrier.hip -o grid_barrier.so"""
import ctypes as C, os, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "grid_barrier.so"))
lib.launch_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
lib.launch_empty.argtypes = [C.c_void_p, C.c_int, C.c_int]
s = torch.cuda.current_stream().cuda_stream
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for blocks in (256, 128, 64):
    for per_block in (64, 1024, 16384):
        phases = 200
        buf = torch.zeros(2 * blocks * per_block, device="cuda")
        cnt = torch.zeros(1, device="cuda", dtype=torch.int32)
        err = torch.zeros(1, device="cuda", dtype=torch.int32)
        sink = torch.zeros(1, device="cuda")
        def go():
            cnt.zero_()
            lib.launch_probe(s, buf.data_ptr(), cnt.data_ptr(), blocks, phases, per_block, err.data_ptr(), 2000000, sink.data_ptr())
        us = timed(go)
        print(f"blocks={blocks} slice={per_block * 4} B: {us / phases:6.2f} us per phase (publish + grid barrier + read-all), err={int(err.item())}", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    lib.launch_empty(torch.cuda.current_stream().cuda_stream, 256, 200)
print(f"200 empty kernels of 256 blocks in a HIP graph: {timed(g.replay) / 200:6.2f} us per kernel boundary")
