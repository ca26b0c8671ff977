"""(r05, negative result: needs the CRAB_NORM_NT switch of the experiment build, see profiles/README.md - kept as the record of the method.)
A/B of the prefill-sized fp32-row norm with and without non-temporal row loads (CRAB_NORM_NT=0 | 1, read once per process: two child
processes), at the two shapes of the prefill phase: decoder RMSNorm [24570, 4096] and CLIP LayerNorm [195320, 1024]; the outputs must be bit-identical.
The rows are rotated through 3 buffers (> the 256 MiB MALL together) so that a launch finds them in HBM, as in the step."""
import hashlib, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import torch
    from crab_amd import ops
    for (M, D, rms) in ((24570, 4096, True), (195320, 1024, False)):
        g = torch.Generator(device="cuda").manual_seed(1)
        xs = [torch.randn(M, D, device="cuda", generator=g) for _ in range(3)]
        w = 1 + 0.1 * torch.randn(D, device="cuda", generator=g)
        b = 0.05 * torch.randn(D, device="cuda", generator=g)
        out = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        run = (lambda x: ops.rmsnorm(x, w, 1e-5, out)) if rms else (lambda x: ops.layernorm(x, w, b, 1e-5, out))
        for i in range(6): run(xs[i % 3])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(60): run(xs[i % 3])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 60 * 1e3
        run(xs[0]); torch.cuda.synchronize()
        h = hashlib.sha1(out.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"NT={os.environ.get('CRAB_NORM_NT', '1')} {'rms' if rms else 'ln '} [{M}, {D}]: {us:7.1f} us  {M * D * 6 / us / 1e6:6.2f} TB/s  sha1 {h}", flush=True)
else:
    for rep in range(2):
        for nt in ("0", "1"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, CRAB_NORM_NT=nt), check=True)
