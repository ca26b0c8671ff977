"""Experiment: does the prefill of the NEXT batch (MFMA-bound) overlap with the decode of the current one (HBM-bound)
when they run on two HIP streams?  usage: overlap_prefill_decode.py [B] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
from crab_amd.build_model import build_crab
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 96
model = build_crab("llama")
um = model.base_model.model
eng = um._engine
D = um.config.hidden_size
g = torch.Generator(device="cuda").manual_seed(0)
embA = (torch.randn(B, 702, D, device="cuda", generator=g) * 0.02).bfloat16()
embB = (torch.randn(B, 702, D, device="cuda", generator=g) * 0.02).bfloat16()
# group A: prefilled decode state + captured graph (engine internals)
st = eng._start(embA, 256, None, 2, 0, 16, False, 0, None)
graph = eng._capture(st)
kcB, vcB = eng.alloc_cache(B, 1024)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def decode(n):
    with torch.cuda.stream(s1):
        for _ in range(n): graph.replay()
def prefill():
    ops.WS_SLOT = 1
    with torch.cuda.stream(s2):
        for b0 in range(0, B, 16):
            eng.prefill(embB[b0:b0 + 16], kcB, vcB, b0=b0)
    ops.WS_SLOT = 0
def timed(fn):
    torch.cuda.synchronize(); t = time.time(); fn(); torch.cuda.synchronize(); return time.time() - t
print('built', flush=True)
prefill(); torch.cuda.synchronize(); print('prefill ok', flush=True)
decode(4); torch.cuda.synchronize(); print('decode ok', flush=True)
td = timed(lambda: decode(STEPS))
tp = timed(prefill)
print('serial', td, tp, flush=True)
tb = timed(lambda: (prefill(), decode(STEPS)))
tb2 = timed(lambda: (decode(STEPS), prefill()))
print(f"B={B}: decode {STEPS} steps {td*1e3:.1f} ms ({td/STEPS*1e3:.2f} ms/step), prefill {tp*1e3:.1f} ms, serial sum {1e3*(td+tp):.1f} ms")
print(f"concurrent (prefill enqueued first) {tb*1e3:.1f} ms, (decode enqueued first) {tb2*1e3:.1f} ms -> overlap saves {(td+tp-min(tb,tb2))*1e3:.1f} ms")
