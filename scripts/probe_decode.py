"""Decode-regime probe (GPU box): B sequences x S = 702 rows, NEW greedy tokens on the decoder only.  Prints ms per decode step
(difference of two generate() lengths) and, under `rocprofv3 --kernel-trace --stats`, gives the per-kernel table of the decode path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd.build_model import build_crab

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NEW = int(sys.argv[2]) if len(sys.argv) > 2 else 64
LLM = sys.argv[3] if len(sys.argv) > 3 else "llama"
G = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # decode groups replayed on separate HIP streams
model = build_crab(LLM, visual=False, audio=False, conditioned=True)
um = model.base_model.model
eng = um._engine
g = torch.Generator(device="cuda").manual_seed(1)
emb = torch.randn(B, 702, um.config.hidden_size, device="cuda", generator=g).to(torch.bfloat16)


def run(n):
    eng.generate(emb, n, eos_token_id=None, pad_token_id=2, decode_streams=G)
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = eng.generate(emb, n, eos_token_id=None, pad_token_id=2, decode_streams=G)
    torch.cuda.synchronize()
    return time.perf_counter() - t, r


t1, _ = run(NEW)
t2, r = run(2 * NEW)
per = (t2 - t1) / NEW
wbytes = sum(p.numel() for p in um.model.layers.parameters()) * 2 + um.lm_head.weight.numel() * 2
print(f"B={B} G={G}: generate({NEW}) {t1*1e3:.1f} ms, generate({2*NEW}) {t2*1e3:.1f} ms -> {per*1e3:.3f} ms/step; weights {wbytes/1e9:.2f} GB -> "
      f"{wbytes/per/1e12:.2f} TB/s equivalent ({wbytes/8e12/per*100:.1f}% of the 8 TB/s floor)", flush=True)
print("ids", r[0, :8].tolist())
