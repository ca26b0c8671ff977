"""debug: where does the 2-3 % first-step logit difference between a 1-clip and a 2-clip call come from? encoders (inputs_embeds) or decoder?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import synth
from crab_amd.build_model import build_crab
model = build_crab(os.environ.get("LLM", "llama"))
um = model.base_model.model
tab = um.SPECIAL_TOKEN_2_IDS
def inputs(B):
    ids = [synth.synth_prompt_ids(128, model.base_vocab, tab, clip=i).cuda() for i in range(B)]
    mods = [{'<video>': synth.synth_video(8, clip=i).cuda(), '<audio>': synth.synth_audio(10, 98, clip=i).cuda()} for i in range(B)]
    return dict(batch_input_ids=ids, batch_labels=[torch.full_like(i, -100) for i in ids], batch_X_modals=mods, batch_task_names=['avqa'] * B)
rel = lambda a, b: float((a.float() - b.float()).abs().max()) / float(b.float().abs().max())
E = {}
for B in (1, 2, 8, 40):
    d = model.prepare_multimodal_inputs(**inputs(B))
    E[B] = d["inputs_embeds"][0].float().clone()
    print(f"B={B}: inputs_embeds of clip 0 vs the 1-clip call: {rel(E[B], E[1]):.5f} of scale; rows that differ: {int(((E[B] - E[1]).abs().amax(-1) > 0).sum())} of {E[B].shape[0]}")
# decoder alone: the SAME embeddings (clip 0 of the 1-clip call), alone and as row 0 of batches whose other rows are other clips' embeddings
d8 = model.prepare_multimodal_inputs(**inputs(8))["inputs_embeds"]
base = d8.clone(); base[0] = E[1].to(base.dtype)
lg = {}
for B in (1, 2, 8):
    out = um(inputs_embeds=base[:B].contiguous())
    lg[B] = out.logits[0, -1].float().clone()
    print(f"decoder only, B={B}: last-row logits of row 0 vs B=1: {rel(lg[B], lg[1]):.5f} of scale (scale {float(lg[1].abs().max()):.2f})")
# sensitivity: the 1-clip embeddings perturbed by bf16-sized noise
for eps in (1e-3, 4e-3):
    x = base[:1].clone().float()
    x = (x + eps * x.abs().max() * torch.randn_like(x) / 3).to(base.dtype)
    o = um(inputs_embeds=x).logits[0, -1].float()
    print(f"decoder sensitivity: embeddings + noise of {eps} of their scale (gaussian / 3) -> logits move {rel(o, lg[1]):.5f} of scale")
