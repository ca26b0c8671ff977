"""Prefill-phase probe (GPU box): encoders + splice and chunked decoder prefill of B clips at the BASELINE AVQA shape.
usage: probe_prefill.py [B] [iters] [enc|pre|both] [chunk]; run under rocprofv3 --kernel-trace --stats for the per-kernel breakdown."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import synth
from crab_amd.build_model import build_crab
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 3
MODE = sys.argv[3] if len(sys.argv) > 3 else "both"      # enc | pre | both
CH = int(sys.argv[4]) if len(sys.argv) > 4 else 16         # sequences per prefill chunk
model = build_crab("llama")
um = model.base_model.model
tab = um.SPECIAL_TOKEN_2_IDS
ids = [synth.synth_prompt_ids(128, model.base_vocab, tab, clip=i).cuda() for i in range(B)]
mods = [{'<video>': synth.synth_video(8, clip=i).cuda(), '<audio>': synth.synth_audio(10, 98, clip=i).cuda()} for i in range(B)]
lab = [torch.full_like(i, -100) for i in ids]
eng = um._engine
kc, vc = eng.alloc_cache(B, 1024)
def enc():
    return um.prepare_multimodal_inputs(ids, lab, mods, ['avqa'] * B)['inputs_embeds']
def pre(emb):
    for b0 in range(0, B, CH):
        eng.prefill(emb[b0:b0 + CH], kc, vc, b0=b0)
emb = enc(); pre(emb); torch.cuda.synchronize()
t0 = time.time()
for _ in range(IT if MODE != 'pre' else 0): emb = enc()
torch.cuda.synchronize(); t1 = time.time()
for _ in range(IT if MODE != 'enc' else 0): pre(emb)
torch.cuda.synchronize(); t2 = time.time()
S = emb.shape[1]
fe = B * (8 * (155.3e9 + 4.0e9) + 10 * (12 * 0.687e9 + 0.453e9 + 0.050e9 + 2.57e9))
fd = B * (S * (2 * 6.476e9 + 90.3e6) + 2 * S * S * 131072)
te, td = (t1 - t0) / IT, (t2 - t1) / IT
print(f"encoders+splice B={B}: {te*1e3:.1f} ms ({te/B*1e3:.2f} ms/clip) -> {fe/te/1e12:.0f} TFLOP/s", flush=True)
print(f"decoder prefill S={S} B={B} chunk={CH}: {td*1e3:.1f} ms ({td/B*1e3:.2f} ms/clip) -> {fd/td/1e12:.0f} TFLOP/s", flush=True)
print(f"phase: {(te+td)/B*1e3:.2f} ms/clip -> {(fe+fd)/(te+td)/1e12:.0f} TFLOP/s = {(fe+fd)/(te+td)/2.5e15*100:.1f}% of 2.5 PF", flush=True)
