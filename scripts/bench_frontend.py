"""Front-end micro-benchmark (GPU box): CLIP preprocessing of uint8 frames resident in HBM and kaldi fbank of 2 s segments."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crab_amd import frontend, synth
def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
proc = frontend.CLIPImageProcessor(dtype=torch.bfloat16)
for (h, w, n) in ((224, 224, 2048), (360, 640, 2048), (1080, 1920, 256)):
    x = torch.randint(0, 256, (n, h, w, 3), dtype=torch.uint8, device="cuda")
    def run():
        r, top, left = proc.resize_crop(x)
        return r
    ms_r = timeit(run)
    frames = [x[i] for i in range(n)]
    ms = timeit(lambda: proc.preprocess(frames), n=3)
    print(f"clip preprocess {n} frames {h}x{w}: resize {ms_r:.2f} ms ({x.numel()/ms_r/1e6:.1f} GB/s in), full call {ms:.2f} ms -> {n/ms*1e3:.0f} frames/s", flush=True)
w = torch.from_numpy(np.stack([synth.synth_waveform(2.0, i) for i in range(16)])).cuda().repeat(160, 1)    # 2560 segments = 256 clips
ms = timeit(lambda: frontend.preprocess(w))
print(f"kaldi fbank {w.shape[0]} x 2 s segments: {ms:.2f} ms -> {w.shape[0]*2/ms*1e3:.0f} s of audio per s, {w.numel()*4/ms/1e6:.1f} GB/s in", flush=True)
