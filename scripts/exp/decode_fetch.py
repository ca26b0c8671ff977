"""Experiment: one eager (no HIP graph) generate() of 256 clips x 4 tokens, to be run under `rocprofv3 --pmc FETCH_SIZE`
so that the decode-regime kernels' L2-side traffic can be compared with their operand bytes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd.build_model import build_crab
model = build_crab("llama", visual=False, audio=False)
um = model.base_model.model
g = torch.Generator(device="cuda").manual_seed(0)
emb = (torch.randn(256, 702, um.config.hidden_size, device="cuda", generator=g) * 0.05).bfloat16()
out = um.generate(inputs_embeds=emb, max_new_tokens=4, min_new_tokens=4, eos_token_id=2, pad_token_id=2, use_graph=False)
torch.cuda.synchronize()
print(out.shape)
