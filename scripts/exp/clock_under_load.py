"""Experiment: shader clock / power while the ring GEMM (random data) and the decode attention run back to back."""
import sys, os, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
BF = torch.bfloat16
x = torch.randn(11232, 4096, device="cuda").to(BF); w = (torch.randn(22016, 4096, device="cuda") * 0.02).to(BF); y = torch.empty(11232, 22016, device="cuda", dtype=BF)
xz = torch.zeros_like(x); wz = torch.zeros_like(w)
B, H, d, Tmax = 256, 32, 128, 960
kc = (torch.randn(B, H, Tmax, d, device="cuda") * 0.5).to(BF); vc = kc.clone(); q = torch.randn(B, H * d, device="cuda").to(BF); o = torch.empty_like(q)
def sample(tag, fn, secs=4.0):
    stop = [False]; out = []
    def poll():
        while not stop[0]:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            sclk = [l for l in r.splitlines() if "sclk" in l]; pw = [l for l in r.splitlines() if "ower" in l and "W" in l]
            out.append((sclk[:1], pw[:1]))
            time.sleep(0.5)
    th = threading.Thread(target=poll); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(20): fn()
        torch.cuda.synchronize(); n += 20
    stop[0] = True; th.join()
    print(tag, f"{(time.time()-t0)/n*1e6:.0f} us/launch"); [print("   ", a, b) for a, b in out[2:6]]
sample("ring GEMM random operands", lambda: ops.gemm(x, w, out=y))
sample("ring GEMM zero operands  ", lambda: ops.gemm(xz, wz, out=y))
sample("decode attention         ", lambda: ops.attn_decode(q, kc, vc, o, B, H, H, d, Tmax, 830, d ** -0.5))
