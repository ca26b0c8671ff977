"""Encode + prefill vs decode time of ONE generate() call at the reference's own batch sizes (1 and 8 clips), full model, 256 new tokens."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import ops, synth
from crab_amd.build_model import build_crab
model = build_crab("llama", device=torch.device("cuda", 0), seed=42)
um = model.base_model.model
tab = um.SPECIAL_TOKEN_2_IDS
for B in (1, 8):
    ids = [synth.synth_prompt_ids(128, model.base_vocab, tab, clip=i).cuda() for i in range(B)]
    mods = [{'<video>': synth.synth_video(8, clip=i).cuda(), '<audio>': synth.synth_audio(10, 98, clip=i).cuda()} for i in range(B)]
    lab = [torch.full_like(i, -100) for i in ids]
    def go():
        return model.generate(batch_input_ids=ids, batch_labels=lab, batch_X_modals=mods, batch_task_names=['avqa'] * B, use_cache=True, max_new_tokens=256,
                              min_new_tokens=256, eos_token_id=um.config.eos_token_id, pad_token_id=um.model.pad_token_id)
    go(); go(); torch.cuda.synchronize()
    prof = ops.KernelProfiler(phase_only=True); ops.PROFILER = prof
    t0 = time.perf_counter(); go(); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    ops.PROFILER = None
    pre, dec = prof.phase_ms()
    print(f"B={B}: wall {wall:.1f} ms = encoders + prefill {pre:.1f} ms + decode {dec:.1f} ms ({dec / 255:.3f} ms per step) + host {wall - pre - dec:.1f} ms", flush=True)
    # split encoders from the decoder prefill: prepare_multimodal_inputs alone
    torch.cuda.synchronize(); t0 = time.perf_counter()
    um.prepare_multimodal_inputs(ids, lab, mods, ['avqa'] * B); torch.cuda.synchronize()
    print(f"      prepare_multimodal_inputs alone (encoders + splice, host-timed): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
