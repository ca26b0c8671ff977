"""Stage timing probe (GPU box): encoders / prefill / decode step at the BASELINE AVQA shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import synth
from crab_amd.build_model import build_crab

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NEW = int(sys.argv[2]) if len(sys.argv) > 2 else 32
t0 = time.time()
model = build_crab("llama")
torch.cuda.synchronize()
print(f"build {time.time()-t0:.1f}s  mem {torch.cuda.memory_allocated()/2**30:.1f} GiB", flush=True)
um = model.base_model.model
tab = um.SPECIAL_TOKEN_2_IDS
ids = [synth.synth_prompt_ids(128, model.base_vocab, tab, clip=i) for i in range(B)]
mods = [{'<video>': synth.synth_video(8, clip=i).cuda(), '<audio>': synth.synth_audio(10, 98, clip=i).cuda()} for i in range(B)]
lab = [torch.full_like(i, -100) for i in ids]

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return (time.time() - t) / n, r

t, inp = timed(lambda: um.prepare_multimodal_inputs(ids, lab, mods, ['avqa'] * B))
emb = inp['inputs_embeds']
print(f"encoders+splice B={B}: {t*1e3:.1f} ms  ({t/B*1e3:.2f} ms/clip)  S={emb.shape[1]}", flush=True)
eng = um._engine
S = emb.shape[1]
kc, vc = eng.alloc_cache(B, 1024)
t, _ = timed(lambda: eng.prefill(emb[:8], kc, vc, b0=0))
fl = 8 * (S * (2 * 6.476e9 + 90.3e6) + 2 * S * S * 131072)
print(f"prefill 8 clips: {t*1e3:.1f} ms -> {fl/t/1e12:.1f} TFLOP/s ({fl/t/2.5e15*100:.1f}% of 2.5PF)", flush=True)
t, out = timed(lambda: um.generate(inputs_embeds=emb, max_new_tokens=NEW, min_new_tokens=NEW, eos_token_id=2, pad_token_id=2), n=2)
print(f"generate B={B} new={NEW}: {t*1e3:.1f} ms", flush=True)
t2, out = timed(lambda: um.generate(inputs_embeds=emb, max_new_tokens=2 * NEW, min_new_tokens=2 * NEW, eos_token_id=2, pad_token_id=2), n=2)
per = (t2 - t) / NEW
print(f"decode step B={B}: {per*1e3:.3f} ms/step -> weights 13.3GB => {13.3e9/per/1e12:.2f} TB/s equiv")
print("ids sample", out[0, :8].tolist())
