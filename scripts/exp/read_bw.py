"""Experiment: achievable read-only HBM bandwidth (torch reductions over a 4 GiB buffer) vs the decode-attention kernel."""
import torch, time
x = torch.empty(1 << 30, device="cuda", dtype=torch.float32).normal_()
def t(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = t(lambda: x.sum())
print(f"torch sum fp32 4 GiB: {ms:.3f} ms -> {x.numel()*4/ms/1e6:.0f} GB/s")
ms = t(lambda: x.max())
print(f"torch max fp32 4 GiB: {ms:.3f} ms -> {x.numel()*4/ms/1e6:.0f} GB/s")
y = torch.empty_like(x)
ms = t(lambda: y.copy_(x))
print(f"torch copy 4 GiB: {ms:.3f} ms -> {2*x.numel()*4/ms/1e6:.0f} GB/s (read+write)")
# ---- hand-written streaming read (scripts/exp/read_bw.hip): blocks x loads-in-flight sweep
import ctypes as C, os
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "read_bw.so"))
lib.launch_read.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_int]
out = torch.zeros(1 << 20, device="cuda", dtype=torch.int32)
nbytes = x.numel() * 4
for nt in (0, 1):
    for blocks in (2048, 4096, 8192, 16384):
        for un in (1, 4, 8):
            fn = lambda: lib.launch_read(torch.cuda.current_stream().cuda_stream, x.data_ptr(), nbytes, out.data_ptr(), blocks, un, nt)
            ms = t(fn)
            print(f"read kernel {'nt     ' if nt else 'default'} blocks={blocks:6d} loads in flight={un}: {nbytes/ms/1e6:.0f} GB/s")
lib.launch_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int]
for ntl in (0, 1):
    for nts in (0, 1):
        for blocks in (2048, 8192):
            fn = lambda: lib.launch_copy(torch.cuda.current_stream().cuda_stream, x.data_ptr(), y.data_ptr(), nbytes, blocks, ntl, nts)
            ms = t(fn)
            print(f"copy kernel nt-load={ntl} nt-store={nts} blocks={blocks:6d}: {2*nbytes/ms/1e6:.0f} GB/s (read+write)")
lib.launch_read_pol.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_int]
names = ["default", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt", "sc0"]
for pol in range(7):
    for blocks in (2048, 8192):
        fn = lambda: lib.launch_read_pol(torch.cuda.current_stream().cuda_stream, x.data_ptr(), nbytes, out.data_ptr(), blocks, pol)
        ms = t(fn)
        print(f"read (asm, 2 loads in flight) policy [{names[pol]:11s}] blocks={blocks:6d}: {nbytes/ms/1e6:.0f} GB/s")
