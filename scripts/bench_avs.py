"""BASELINE configs[4] on one MI355X: the AVSS pixel-level path at full size - Llama-2-7B hyper-LoRA decoder (32 layers) + CLIP ViT-L/14 (multi-scale
features) + BEATs + both Q-Formers + SegModule - one sample per generate_avs call as the reference's loops run it (scripts/quick_start.py:361-450:
batch of one, max_new_tokens = 100), for the binary head (s4 / ms3 / ref-avs: 1 class plane) and the 71-class AVSS head, followed by what the loop
does with the masks: label map for the PNG (crab_mask_labels), metrics (crab_amd.avss_utils), all on the device.

Synthetic: seeded N(0, 0.02) weights, one 224 x 224 image, one 1-s audio window, a 48-token prompt.  A random decoder never emits the six <mask_i>
tokens, so their ids are re-pointed at tokens it does emit (tests/test_fullsize_gpu.py does the same): the selection + SegModule work is what the
reference does for a sample that segments.  Prints one JSON object; `python scripts/bench_avs.py > profiles/<name>.json`."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from crab_amd import avss_utils, ops, synth
from crab_amd.build_model import build_crab

NEW = int(os.environ.get("CRAB_AVS_NEW_TOKENS", "100"))


def wall(fn, n=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


t0 = time.perf_counter()
model = build_crab("llama", segment=True, seed=5)
g = torch.Generator(device="cuda").manual_seed(77)
for name, buf in model.named_buffers():                # the SAM-style random Fourier matrices are buffers
    if name.endswith("positional_encoding_gaussian_matrix"):
        buf.normal_(generator=g)
build_s = time.perf_counter() - t0
sp = model.SPECIAL_TOKEN_2_IDS
ids = synth.synth_prompt_ids(48, model.base_vocab, sp, clip=3)
for a_, b_ in (("<video_start>", "<image_start>"), ("<video>", "<image>"), ("<video_end>", "<image_end>")):
    ids[ids == sp[a_]] = sp[b_]
mods = [{'<image>': synth.synth_video(1, clip=3).cuda(), '<audio>': synth.synth_audio(1, 98, clip=3).cuda()}]
lab = [torch.full_like(ids, -100)]
out = {"config": "BASELINE configs[4]: AVSS pixel-level path, Llama-2-7B + CLIP ViT-L/14 + BEATs + SegModule, bf16, 1 x MI355X, one sample per call",
       "new_tokens": NEW, "prompt_tokens": int(ids.numel()), "build_s": round(build_s, 1), "data": "synthetic (seeded weights, one image, one audio window)"}
for task in ("s4", "avss"):
    kw = dict(batch_input_ids=[ids.cuda()], batch_labels=lab, batch_X_modals=mods, batch_task_names=[task], max_new_tokens=NEW, pad_token_id=2, eos_token_id=None)
    plain = model.generate(**kw).cpu()
    for i in range(6):
        sp[f'<mask_{i}>'] = int(plain[0, NEW - 7 + i])      # the six picks sit at the end of the generation, as a trained model's do
    ms_gen, _ = wall(lambda: model.generate(**kw))
    ms_avs, res = wall(lambda: model.generate_avs(**kw))
    assert res["pred_masks"] is not None and torch.equal(res["output_ids"].cpu(), plain)
    pred = res["pred_masks"][0].float()
    C = pred.shape[0]
    if C == 1:
        gt = (torch.rand(1, 224, 224, device="cuda") > 0.5).float()
        ms_metric, vals = wall(lambda: (avss_utils.mask_iou(pred, gt).item(), avss_utils.Eval_Fmeasure(pred, gt)), n=10)
    else:
        gt = torch.randint(0, C, (1, 224, 224), device="cuda")
        ms_metric, vals = wall(lambda: [v.sum().item() for v in avss_utils.calc_color_miou_fscore(pred.unsqueeze(0), gt, T=1)[:3]], n=10)
    ms_lab, _ = wall(lambda: ops.mask_labels(pred).cpu(), n=10)
    out[task] = {"num_classes": int(C), "generate_ms": round(ms_gen, 1), "generate_avs_ms": round(ms_avs, 1),
                 "mask_path_ms": round(ms_avs - ms_gen, 1), "samples_per_s": round(1e3 / ms_avs, 3),
                 "ms_per_token": round(ms_gen / NEW, 3), "metrics_ms_values_read_back": round(ms_metric, 3), "png_labels_ms_incl_copy": round(ms_lab, 3),
                 "note": "generate_avs = generate with per-step hidden states + the picked states through SegModule (two mask-decoder levels, 300 queries, "
                         "bilinear 112 -> 224); mask_path_ms = what the pixel head adds to plain generation"}
print(json.dumps(out))
