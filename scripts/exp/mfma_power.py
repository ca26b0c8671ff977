"""Sustained register-only bf16 MFMA rate by instruction shape and operand data (see mfma_power.hip).  Each configuration runs ~3 s of back-to-back
launches; the rate of the first and of the last 0.5 s are printed (the power cap pulls the clock down within the first second)."""
import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "mfma_power.so"))
lib.launch_mfma.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
sink = torch.zeros(4, device="cuda")
BLOCKS, ITERS = 256, 40000
flops = BLOCKS * 8 * ITERS * 2 * 32 * 16384.0


def power():
    try:
        out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=10).stdout
        return " | ".join(l.strip() for l in out.splitlines() if "Power" in l)[:160]
    except Exception as e:      # noqa: BLE001
        return f"(rocm-smi: {e})"


for shape, zero in ((16, 0), (32, 0), (16, 1), (32, 1), (16, 0), (32, 0)):
    s = torch.cuda.current_stream().cuda_stream
    lib.launch_mfma(s, shape, sink.data_ptr(), BLOCKS, 100, zero); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(61)]
    ev[0].record()
    for i in range(60):
        lib.launch_mfma(s, shape, sink.data_ptr(), BLOCKS, ITERS, zero)
        ev[i + 1].record()
        if i == 50:
            ev[50].synchronize(); pw = power()
    torch.cuda.synchronize()
    ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(60)]
    first, last = sum(ts[:5]) / 5, sum(ts[-10:]) / 10
    print(f"{shape}x{shape} {'zero  ' if zero else 'random'}: first launches {flops / first / 1e9:7.1f} TFLOP/s, sustained {flops / last / 1e9:7.1f} TFLOP/s "
          f"({last:.1f} ms per launch; total {sum(ts) / 1e3:.1f} s)  {pw}", flush=True)
    torch.cuda.synchronize()
    import time; time.sleep(2.0)
