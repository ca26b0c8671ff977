"""Segmentation-metrics micro-benchmark (GPU box): the eval loops' per-sample scoring at the eval shapes - one 224 x 224 binary mask
(mask_iou + Eval_Fmeasure), one 71-class 224 x 224 AVSS frame (calc_color_miou_fscore) - and the same at 64 frames per call, on masks
resident in HBM; next to it the host path the reference runs (its functions restated in oracle/metrics_oracle.py; the reference's own
torch-CPU Eval_Fmeasure sweeps the image 255 times) on this box's cores, INCLUDING the device-to-host copy its `.cpu()` pays.
Prints launch time by HIP events and the HBM rate of the counting pass (algorithmic bytes: every mask plane read once)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from crab_amd import _lib, avss_utils as AU
from crab_amd.ops import _p, _stream
from oracle import metrics_oracle as MO


def timeit(fn, n=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3          # us


def host(fn, n=3):
    fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t) * 1e6


lib, ctx = _lib.load(), _lib.ctx(0)
H = W = 224
for N in (1, 64):
    pred = torch.randn(N, H, W, device="cuda") * 3
    gt = (torch.rand(N, H, W, device="cuda") > 0.6).float()
    counts = torch.empty(N, 6, device="cuda", dtype=torch.int32)
    out = torch.empty(2, device="cuda")
    us = timeit(lambda: lib.crab_mask_iou(ctx, _stream(), _p(pred), _p(gt), N, H * W, 1e-7, _p(counts), _p(out)))
    byt = 2 * N * H * W * 4
    print(f"mask_iou        N={N:3d}: {us:8.1f} us per call (memset + count + finish)  {byt / us / 1e3:8.1f} GB/s", flush=True)
    th = AU._thresholds(pred.device, 255)
    ge, ys = torch.empty(N, 2, 255, device="cuda", dtype=torch.int32), torch.empty(N, 2, device="cuda", dtype=torch.int32)
    fs, sc, best = torch.empty(N, 255, device="cuda"), torch.empty(255, device="cuda"), torch.empty(2, device="cuda")
    us = timeit(lambda: lib.crab_fmeasure(ctx, _stream(), _p(pred), _p(gt), N, H * W, _p(th), 255, 0.3, _p(ge), _p(ys), _p(fs), _p(sc), _p(best)))
    print(f"Eval_Fmeasure   N={N:3d}: {us:8.1f} us per call (histogram pass + per-image finish + mean)  {byt / us / 1e3:8.1f} GB/s", flush=True)
    wall = host(lambda: (AU.mask_iou(pred, gt).item(), AU.Eval_Fmeasure(pred, gt)))
    print(f"  both through crab_amd.avss_utils, values read back: {wall:9.1f} us wall", flush=True)
    pc, gc = pred.cpu().numpy(), gt.cpu().numpy()
    hus = host(lambda: (pred.cpu(), gt.cpu(), MO.mask_iou(pc, gc), MO.eval_fmeasure(pc, gc)), n=2 if N > 1 else 3)
    print(f"  host restatement (copy + numpy, one core): {hus:9.1f} us", flush=True)
    C = 71
    cp = torch.randn(N, C, H, W, device="cuda")
    ct = torch.randint(0, C, (N, H, W), device="cuda")
    e = lambda *s, dt=torch.float32: torch.empty(s, device="cuda", dtype=dt)
    areas, iou_fc, a, b, c_, v = e(N, 3, C, dt=torch.int32), e(N, C), e(C), e(C), e(C), e(N)
    us = timeit(lambda: lib.crab_miou_fscore(ctx, _stream(), _p(cp), _p(ct), N, C, H * W, 0.3, _p(areas), _p(iou_fc), _p(a), _p(b), _p(c_), _p(v)))
    byt = N * H * W * (C * 4 + 8)
    print(f"miou_fscore     BF={N:3d} C=71: {us:8.1f} us per call  {byt / us / 1e3:8.1f} GB/s (class planes read once)", flush=True)
    if N == 1:
        cpc, ctc = cp.cpu().numpy(), ct.cpu().numpy()
        hus = host(lambda: (cp.cpu(), MO.batch_miou_fscore(cpc, ctc)))
        print(f"  host restatement (copy + numpy, one core): {hus:9.1f} us", flush=True)
