"""Eval harness around the hot path: what the reference's dataset / collator / inference loop do on
the host between a media file and `UnifiedForCausalLM.generate`, minus the file decoding (decord / librosa / PIL.open stay
with the caller, who hands over uint8 frames and a 16 kHz mono waveform).

  reference                                                              here
  UnifiedTestDataset.add_custome_test_samples  quick_start_dataset.py:149-270   build_instruction
  UnifiedTestDataset.__getitem__               quick_start_dataset.py:277-620   wrap_prompt, frame_indices, audio_windows, make_instance
  DataCollatorForUnifiedTestDataset            quick_start_dataset.py:623-707   Collator
  prepare_sample                               utils/util.py:33-47              to_device
  inference_ntp / inference_avqa               quick_start.py:30-50, inference_hyper_lora.py:158-212   run_inference
  inference_ms3 / _s4 / _avss / _ref_avs       quick_start.py:270-450 (generate_avs -> mask -> PNG + record)   run_inference_avs
  mask_iou / Eval_Fmeasure / metric_s_for_null / calc_color_miou_fscore + the loops' closing averages   utils/avss_utils.py, quick_start.py:118-135, 342-358, 395-447   crab_amd.avss_utils (device), summarise_avs

Image and audio preprocessing run on the device through crab_amd.frontend (HIP kernels); prompts and ids are host work.
The reference loops clip by clip on one GPU; run_inference shards the batches over ranks (crab_amd.parallel) and rank 0
collects the predictions.  Pinned by tests/golden/harness.npz (made by running the reference's dataset + collator)."""
from __future__ import annotations

import json
from typing import Any, Callable, Dict, Iterable, List, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

SYSTEM_PROMPT = "You are a helpful assistant."                      # quick_start_dataset.py:286
_VIDEO_AUDIO = "This is a video:\n<video_start><video><video_end>\nThis is an audio:\n<audio_start><audio><audio_end>\n"
_IMAGE_AUDIO = "This is an image:\n<image_start><image><image_end>\nThis is an audio:\n<audio_start><audio><audio_end>\n"
_SEGMENT = "Please segment out the object that makes the sound in the image."
# task -> (modality header, request); {question} / {exp} are the per-sample fields (quick_start_dataset.py:159-262)
TASK_PROMPTS: Dict[str, Tuple[str, str]] = {
    "avqa": (_VIDEO_AUDIO, "Please answer this question: {question}"),
    "ave": (_VIDEO_AUDIO, "Please describe the events and time range that occurred in the video."),
    "avvp": (_VIDEO_AUDIO, "Please determine the events that occur based on the visual and audio information, as well as the start and "
                           "end time of these events."),
    "arig": (_IMAGE_AUDIO, "Please output the location coordinates of sounding object."),
    "s4": (_IMAGE_AUDIO, _SEGMENT),
    "ms3": (_IMAGE_AUDIO, _SEGMENT),
    "avss": (_IMAGE_AUDIO, _SEGMENT),
    "ref-avs": (_IMAGE_AUDIO, "Please segment out {exp} in the image."),
}
VIDEO_TASKS = ("avqa", "ave", "avvp")
# task -> (clip seconds the waveform is divided into, windows taken): quick_start_dataset.py:318-336 (avqa), :369-385 (ave),
# :409-424 (avvp), :446-450 (s4), :478-482 (ms3), :511-516 (avss), :546-555 (arig), :572-588 (ref-avs)
_AUDIO_PLAN = {"avqa": (60, "two_second"), "ave": (10, "every_second"), "avvp": (10, "every_second"), "ref-avs": (10, "every_second"),
               "s4": (5, "one"), "ms3": (5, "one"), "avss": (10, "one"), "arig": (5, "one_padded")}


def build_instruction(task: str, question: Optional[str] = None, exp: Optional[str] = None) -> str:
    """The instruction string of one sample; `exp` (ref-avs) is lower-cased like the reference does (:255)."""
    if task not in TASK_PROMPTS:
        raise ValueError("invalid task.")
    head, req = TASK_PROMPTS[task]
    if task == "avqa":
        if question is None:
            raise ValueError("avqa needs a question")
        req = req.format(question=question)
    elif task == "ref-avs":
        if exp is None:
            raise ValueError("ref-avs needs a referring expression")
        req = req.format(exp=exp.lower())
    return head + req


def wrap_prompt(tokenizer, instruction: str, output: str = "none") -> Tuple[str, str]:
    """quick_start_dataset.py:284-290: every tokenizer with `apply_chat_template` gets the system + user conversation rendered
    with the generation prompt (Llama-2: `<s>[INST] <<SYS>>...`), and the target gains a literal '</s>'."""
    if tokenizer is not None and hasattr(tokenizer, "apply_chat_template"):
        messages = [{"role": "system", "content": SYSTEM_PROMPT}, {"role": "user", "content": instruction}]
        instruction = tokenizer.apply_chat_template(conversation=messages, add_generation_prompt=True, tokenize=False)
        output = output + "</s>"
    return instruction, output


def encode_text(tokenizer, text: str) -> List[int]:
    """tokenize + convert_tokens_to_ids (collator :661-662): no BOS / EOS of its own, whatever `<s>` there is comes from the text."""
    return tokenizer.convert_tokens_to_ids(tokenizer.tokenize(text))


def frame_indices(vlen: int, n_frames: int) -> List[int]:
    """Uniform frame sampling of :305-309: np.arange(0, vlen, vlen / min(n, vlen)) truncated to int."""
    n = min(int(n_frames), int(vlen))
    if n <= 0:
        raise ValueError("empty video")
    return np.arange(0, vlen, vlen / n).astype(int).tolist()


def audio_windows(task: str, audio, idx: Optional[int] = None) -> torch.Tensor:
    """Waveform [L] (16 kHz mono; numpy or torch) -> the windows the task feeds to the fbank, float32 [T, n].
    avqa: ten 2 s windows centred at 0.5, 6.5, ..., 54.5 s of a 60 s clip, silence-padded at both ends; ave / avvp / ref-avs:
    the ten seconds of a 10 s clip (short tail padded); s4 / ms3 / arig: second `idx` of 5; avss: second `idx` of 10.
    (The reference's padded windows become float64 through np.zeros(dtype=float); everything is float32 here.)"""
    if task not in _AUDIO_PLAN:
        raise ValueError("invalid task.")
    a = torch.as_tensor(np.asarray(audio) if not isinstance(audio, torch.Tensor) else audio).to(torch.float32).reshape(-1)
    tot, kind = _AUDIO_PLAN[task]
    nps = int(a.shape[0] / tot)
    if kind == "two_second":
        from .frontend import avqa_audio_segments
        return avqa_audio_segments(a, tot)
    if kind == "every_second":
        segs = []
        for i in range(tot):
            s = a[int(max(0, i) * nps): int(nps * min(tot, i + 1))]
            if s.shape[0] < nps:
                s = torch.cat([s, s.new_zeros(nps - s.shape[0])])
            segs.append(s)
        return torch.stack(segs, 0)
    if idx is None:
        raise ValueError(f"{task} needs the frame index of the sample")
    s = a[idx * nps: (idx + 1) * nps]
    if kind == "one_padded" and s.shape[0] < nps:
        s = torch.cat([s, s.new_zeros(nps - s.shape[0])])
    return s[None]


def make_instance(task: str, tokenizer, *, question: Optional[str] = None, exp: Optional[str] = None, frames: Optional[Sequence] = None,
                  image=None, audio=None, idx: Optional[int] = None, mask: Optional[torch.Tensor] = None, n_frames: int = 10,
                  processor=None, output: str = "none", device="cuda") -> Dict[str, Any]:
    """One dataset item (`UnifiedTestDataset.__getitem__`): wrapped instruction, target, task name and the preprocessed
    modalities.  frames: the decoded video as a sequence of uint8 [H,W,3] images (all of them - sampling happens here);
    image: one uint8 [H,W,3]; audio: waveform [L].  Preprocessing runs on `device` (crab_amd.frontend)."""
    from . import frontend
    instruction, output = wrap_prompt(tokenizer, build_instruction(task, question, exp), output)
    data: Dict[str, Any] = {"instruction": instruction, "output": output, "task_name": task}
    proc = processor if processor is not None else frontend.CLIPImageProcessor(device=device)
    if task in VIDEO_TASKS:
        if frames is None or audio is None:
            raise ValueError(f"{task} needs frames and audio")
        pick = frame_indices(len(frames), n_frames)
        data["video"] = proc.preprocess([frames[i] for i in pick], return_tensors="pt")["pixel_values"]      # [T,3,224,224]
    else:
        if image is None or audio is None:
            raise ValueError(f"{task} needs an image and audio")
        data["image"] = proc.preprocess([proc.resize_exact(image, 224, 224)], return_tensors="pt")["pixel_values"]   # .resize((224,224)) first (:456)
        if mask is not None:
            data["mask"] = mask
    win = audio_windows(task, audio, idx).to(device)
    fb = frontend.preprocess(win).to(torch.float32)                                                         # [T, L_a, 128]
    data["audio"] = fb if _AUDIO_PLAN[task][1] in ("two_second", "every_second") else fb[0]
    return data


class Collator:
    """DataCollatorForUnifiedTestDataset (:623-707): prompts -> ids (labels all -100), modalities keyed by placeholder."""

    _KEYS = (("image", "<image>"), ("video", "<video>"), ("audio", "<audio>"), ("mask", "<mask>"))

    def __init__(self, tokenizer):
        self.tokenizer = tokenizer

    def __call__(self, instances: Sequence[Mapping[str, Any]]) -> Dict[str, list]:
        out = {"batch_input_ids": [], "batch_labels": [], "batch_X_modals": [], "batch_metadata": [], "batch_task_names": []}
        for inst in instances:
            ids = encode_text(self.tokenizer, inst["instruction"])
            out["batch_input_ids"].append(torch.tensor(ids, dtype=torch.long))
            out["batch_labels"].append(torch.full((len(ids),), -100, dtype=torch.long))
            meta = {"instruction": inst["instruction"], "output": inst["output"]}
            mods = {}
            for key, tag in self._KEYS:
                if inst.get(key) is not None:
                    mods[tag] = inst[key]
                    meta[key + "_path"] = inst.get(key + "_path", "")
            out["batch_X_modals"].append(mods)
            out["batch_metadata"].append(meta)
            out["batch_task_names"].append(inst["task_name"])
        return out


def to_device(data, device="cuda"):
    """utils/util.py:33-47 `prepare_sample`: tensors inside nested dicts / lists / tuples move to the device."""
    if isinstance(data, Mapping):
        return type(data)({k: to_device(v, device) for k, v in data.items()})
    if isinstance(data, (tuple, list)):
        return type(data)(to_device(v, device) for v in data)
    if isinstance(data, torch.Tensor):
        return data.to(device)
    return data


def run_inference(batches: Iterable[Mapping[str, Any]], model, tokenizer, max_new_tokens: int = 500, out_path: Optional[str] = None,
                  device="cuda", rank: int = 0, world: int = 1, on_result: Optional[Callable[[dict], None]] = None, in_flight: int = 1,
                  coalesce: bool = False, coalesce_rows: int = 448, **generate_kwargs) -> List[dict]:
    """inference_ntp / inference_avqa: for every collated batch, generate -> batch_decode(skip_special_tokens=False) ->
    metadata['predict'], appended to `out_path` as JSON lines when given.  With world > 1 batch i runs on rank i mod world and
    rank 0 receives every record (returned in batch order; other ranks return their own records).
    in_flight > 1: that many of this rank's batches decode TOGETHER (model.generate_batches: every batch keeps its own
    prepare_multimodal_inputs / left padding and gets the ids a separate generate() call returns; at the reference's batch of 8 a
    decode step is latency-bound, so three batches in flight finish in ~1.9x the time of one).
    coalesce = True: this rank's batches are collected until they hold `coalesce_rows` clips (or `in_flight` batches when that is given
    > 1) and then decode as ONE ragged batch (model.generate_batches(coalesce=True)): every batch keeps its own prepare_multimodal_inputs
    result, left padding and positions, but a decode step streams the weights once for all of them and the encoders see the clips of all
    batches together - the eval loop's batches of 8 run at the throughput of one large generate()."""
    mine: List[Tuple[int, dict]] = []
    pending: List[Tuple[int, list, dict]] = []

    def flush():
        if not pending:
            return
        kw = {"use_cache": True, "max_new_tokens": max_new_tokens}
        kw.update(generate_kwargs)
        with torch.no_grad():
            if len(pending) == 1:
                ids_list = [model.generate(**pending[0][2], **kw)]
            elif coalesce:
                ids_list = model.generate_batches([s for _, _, s in pending], coalesce=True, max_rows=coalesce_rows, **kw)
            else:
                ids_list = model.generate_batches([s for _, _, s in pending], **kw)
        for (step_, metas_, _), ids in zip(pending, ids_list):
            texts = tokenizer.batch_decode(ids, skip_special_tokens=False)
            for meta, text in zip(metas_, texts):
                rec = dict(meta)
                rec["predict"] = text
                mine.append((step_, rec))
        pending.clear()

    for step, sample in enumerate(batches):
        if step % world != rank:
            continue
        sample = dict(sample)
        metas = sample.pop("batch_metadata")
        pending.append((step, metas, to_device(sample, device)))
        if coalesce and in_flight <= 1:
            if sum(len(p[2]["batch_input_ids"]) for p in pending) >= coalesce_rows:
                flush()
        elif len(pending) >= max(1, in_flight):
            flush()
    flush()
    records = mine
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        if rank == 0:
            records = sorted((r for part in gathered for r in part), key=lambda t: t[0])
    out = [r for _, r in records]
    if rank == 0:
        for rec in out:
            if on_result is not None:
                on_result(rec)
        if out_path is not None:
            with open(out_path, "a") as f:
                for rec in out:
                    f.write(json.dumps(rec) + "\n")
    return out


def default_palette(n: int = 71) -> np.ndarray:
    """The [n, 3] uint8 colour table of the AVSS class map: the PASCAL-VOC bit-shuffle palette (background = class 0 = black).  This IS the
    reference's table: get_v2_pallete (dataset/quick_start_dataset.py:35-59) builds it with the same bit shuffle and reads its cluster file
    label2idx.json only to assert the class count (pinned by tests/golden/seg_metrics.npz: `v2_pallete`)."""
    pal = np.zeros((n, 3), np.uint8)
    for i in range(n):
        c, r, g, b = i, 0, 0, 0
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        pal[i] = (r, g, b)
    return pal


def summarise_avs(records: Sequence[Mapping[str, Any]]) -> Dict[str, Any]:
    """The closing lines of the reference's pixel-task loops over the per-sample records of run_inference_avs (in record order):
      binary tasks   miou = sum(iou) / count (an fp32 tensor sum there), f-score = sum(fscore) / count (Python floats)   quick_start.py:120-135, 200-206
      null refs      ms = sum(s) / count                                                                                 quick_start.py:343-358
      avss           per-class IoU / F sums over the samples divided by the class counts, NaN -> 0, mean over the classes with and without
                     the last one                                                                                        quick_start.py:399-447
    Only the keys whose inputs occur in the records are present."""
    out: Dict[str, Any] = {}
    ious = [r["iou"] for r in records if r.get("iou") is not None]
    if ious:
        acc = np.float32(0)
        for v in ious:
            acc = np.float32(acc + np.float32(v))
        out.update(miou=float(np.float32(acc / np.float32(len(ious)))), f_score=sum(r["fscore"] for r in records if r.get("iou") is not None) / len(ious),
                   count=len(ious))
    ss = [r["s"] for r in records if r.get("s") is not None]
    if ss:
        out.update(ms=sum(ss) / len(ss), count_null=len(ss))
    av = [r["_avss"] for r in records if r.get("_avss") is not None]
    if av:
        sums = [np.zeros(len(av[0][0]), np.float32) for _ in range(3)]
        for triple in av:
            for acc, v in zip(sums, triple):
                acc += np.asarray(v, np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            mi, fs = sums[0] / sums[2], sums[1] / sums[2]
        mi[np.isnan(mi)] = 0
        fs[np.isnan(fs)] = 0
        out["avss"] = {"miou": float(mi.mean(dtype=np.float32)), "miou_noBg": float(mi[:-1].mean(dtype=np.float32)),
                       "f_score": float(fs.mean(dtype=np.float32)), "f_score_noBg": float(fs[:-1].mean(dtype=np.float32)), "count": len(av)}
    return out


get_v2_pallete = default_palette                                    # the reference's name for it (its argument is the class count here)


def color_mask_to_label(mask, v_pallete: Optional[np.ndarray] = None, device="cuda") -> torch.Tensor:
    """dataset/quick_start_dataset.py:63-73 (called at :537 on the mask after `.convert('RGB').resize((224, 224), NEAREST)`): the AVSS colour map
    -> class ids, on the device (crab_color_to_label): `mask` = a PIL RGB image / uint8 array / uint8 tensor [H, W, 3] -> int64 [H, W] = the first
    palette index whose colour equals the pixel, 0 where none does.  `.unsqueeze(0)` of it is the '<mask>' of the avss sample (:538)."""
    from . import _lib
    from .ops import _dev, _p, _stream
    pal = default_palette() if v_pallete is None else np.asarray(v_pallete)
    if pal.ndim != 2 or pal.shape[1] != 3 or not 1 <= pal.shape[0] <= 256 or pal.min() < 0 or pal.max() > 255:
        raise _lib.CrabHipError("color_mask_to_label: palette [n <= 256, 3] with entries in 0..255 expected")
    m = mask if isinstance(mask, torch.Tensor) else torch.from_numpy(np.array(mask))          # (a copy: np.asarray of a PIL image is read-only)
    if m.dtype != torch.uint8 or m.dim() != 3 or m.shape[2] != 3:
        raise _lib.CrabHipError(f"color_mask_to_label: uint8 [H, W, 3] expected, got {m.dtype} {tuple(m.shape)}")
    m = m.to(device).contiguous()
    d = _dev(m)
    p = torch.from_numpy(np.ascontiguousarray(pal.astype(np.uint8))).to(m.device)
    out = torch.empty(m.shape[:2], device=m.device, dtype=torch.int64)
    _lib.check(_lib.load().crab_color_to_label(_lib.ctx(d), _stream(), _p(m), m.shape[0] * m.shape[1], _p(p), int(pal.shape[0]), _p(out)), d)
    return out


def run_inference_avs(batches: Iterable[Mapping[str, Any]], model, tokenizer, out_dir: str, max_new_tokens: int = 100, device="cuda",
                      rank: int = 0, world: int = 1, palette: Optional[np.ndarray] = None, out_path: Optional[str] = None,
                      on_result: Optional[Callable[[dict], None]] = None, metrics: bool = True, null_reference: bool = False,
                      summary: Optional[dict] = None, coalesce: bool = False, coalesce_rows: int = 128, write_png: bool = True,
                      **generate_kwargs) -> List[dict]:
    """The pixel-task loops of the reference (scripts/quick_start.py:270-359 inference_ms3 / _s4 / _ref_avs, :361-450 inference_avss): for every
    collated batch (one sample per batch, like the reference, which reads batch_metadata[0]) generate_avs -> text + masks -> files:
      binary tasks (one class plane):  `<out_dir>/mask_img_dir/<video>/<frame>_pred.png`, mode 'P', 255 where sigmoid(pred) > 0.5 (:313-319)
      avss (71 class planes):          `<out_dir>/avss_result/<video>/<frame>_pred.png`, RGB, palette[argmax over classes] (avss_utils.py:281-312)
    <video> / <frame> come from metadata['mask_path'] (`.../<video>/<fid>/<frame>.png`, :308-311) when present, else from the batch index.
    The thresholding / argmax runs on the device (crab_mask_labels); PNG encoding is host work (Pillow).  A sample whose generation did not
    produce the six <mask_i> tokens has no masks: its record carries pred_path None, as the reference skips it (:303-306).
    Metrics (`metrics=True`, a sample whose X_modals carry the ground truth '<mask>', quick_start_dataset.py:692-694): computed on the DEVICE
    from the predicted mask where the reference first moves it to the host (crab_amd.avss_utils = utils/avss_utils.py): binary tasks get
    'iou' + 'fscore' (mask_iou, Eval_Fmeasure; quick_start.py:118-119, 198-199, 267-268), with `null_reference=True` (the loop over the null
    split, :270-358) 's' instead (metric_s_for_null, :342); avss gets the per-class IoU / F sums of calc_color_miou_fscore (:395-403).
    The ground truth is written next to the prediction as `<frame>_gt.png` ('gt_path'); the resized copy of the input image the reference also
    saves (`_image.jpg`, :111-115) needs the image file and stays with the caller.
    `summary` (a dict, filled on rank 0) receives summarise_avs(records) = the loops' closing averages.
    Batch i runs on rank i mod world; rank 0 returns every record in batch order and appends them to `out_path` as JSON lines when given.
    coalesce = True (r06; BASELINE configs[4] as a throughput path): this rank's samples are collected until `coalesce_rows` of them are pending
    and then run as ONE model.generate_avs_many call - every sample keeps the semantics of its own generate_avs call (own prompt, positions,
    mask-token picks), but the decoder streams its weights once per step for all of them and the SegModule runs batched; the records, files
    and metrics are those of the one-by-one loop (masks agree with it within the mask decoder's bf16 tolerance: different GEMM tile shapes).
    write_png = False skips the PNG encoding / file writes (host work of the reference's loops; the benchmark times the device path)."""
    import os
    from PIL import Image
    from . import ops
    pal = default_palette() if palette is None else np.asarray(palette, np.uint8)
    mine: List[Tuple[int, dict]] = []
    pending: List[tuple] = []
    kw = {"use_cache": True, "max_new_tokens": max_new_tokens}
    kw.update(generate_kwargs)

    def finish(step, meta, task, gt, result):
        rec = {"instruction": meta.get("instruction"), "label": meta.get("output"), "image_path": meta.get("image_path"),
               "predict": tokenizer.decode(result["output_ids"][0], skip_special_tokens=False), "pred_path": None}
        masks = result.get("pred_masks")
        if masks is not None:
            pred = masks[0].float()                                     # [num_classes, 224, 224]
            parts = (meta.get("mask_path") or "").split("/")
            video = parts[-3] if len(parts) >= 3 else f"sample_{step:06d}"
            frame = os.path.splitext(parts[-1])[0] if parts[-1:] and parts[-1] else "0"
            d = os.path.join(out_dir, "mask_img_dir" if pred.shape[0] == 1 else "avss_result", video)
            if write_png:
                lab = ops.mask_labels(pred).cpu().numpy()               # uint8 [224, 224]: 0 / 255, or the class index
                img = Image.fromarray(lab).convert("P") if pred.shape[0] == 1 else Image.fromarray(pal[np.minimum(lab, len(pal) - 1)])
                os.makedirs(d, exist_ok=True)
                rec["pred_path"] = os.path.join(d, frame + "_pred.png")
                img.save(rec["pred_path"], format="PNG")
            rec["task"], rec["num_classes"] = task, int(pred.shape[0])
            if gt is not None:
                from . import avss_utils
                g = gt.to(pred.device)
                # the ground truth beside the prediction, as the reference's loops write it: `<frame>_gt.png`, (sigmoid(gt) > 0.5) * 255 in mode 'P'
                # (quick_start.py:104-109) or palette[gt] with ids outside the table left black (avss_utils.py:315-345 save_gt_mask)
                if not write_png:
                    gimg = None
                elif pred.shape[0] == 1:
                    gimg = Image.fromarray(ops.mask_labels(g.float().reshape(1, *g.shape[-2:]).contiguous()).cpu().numpy()).convert("P")
                else:
                    gl = g.reshape(g.shape[-2:]).cpu().numpy()
                    gimg = Image.fromarray(np.where(((gl >= 0) & (gl < len(pal)))[..., None], pal[np.clip(gl, 0, len(pal) - 1)], 0).astype(np.uint8))
                if write_png:
                    rec["gt_path"] = os.path.join(d, frame + "_gt.png")
                    gimg.save(rec["gt_path"], format="PNG")
                if pred.shape[0] > 1:
                    i_pc, f_pc, c_pc, _ = avss_utils.calc_color_miou_fscore(pred=pred.unsqueeze(0), target=g, T=1)
                    rec["_avss"] = [i_pc.tolist(), f_pc.tolist(), c_pc.tolist()]
                elif null_reference:
                    rec["s"] = avss_utils.metric_s_for_null(pred).item()
                else:
                    rec["iou"] = avss_utils.mask_iou(pred=pred, target=g).item()
                    rec["fscore"] = avss_utils.Eval_Fmeasure(pred=pred, gt=g)
        mine.append((step, rec))

    def flush():
        if not pending:
            return
        with torch.no_grad():
            if len(pending) == 1:
                results = [model.generate_avs(**pending[0][4], **kw)]
            else:
                results = model.generate_avs_many([p[4] for p in pending], **kw)
        for (step_, meta_, task_, gt_, _), result in zip(pending, results):
            finish(step_, meta_, task_, gt_, result)
        pending.clear()

    for step, sample in enumerate(batches):
        if step % world != rank:
            continue
        sample = dict(sample)
        meta = dict(sample.pop("batch_metadata")[0])
        task = sample["batch_task_names"][0]
        gt = sample["batch_X_modals"][0].get("<mask>") if metrics else None            # [1, 224, 224]: {0, 1} fp32, or class ids (avss)
        pending.append((step, meta, task, gt, to_device(sample, device)))
        if not coalesce or len(pending) >= max(1, coalesce_rows):
            flush()
    flush()
    records = mine
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(mine, gathered, dst=0)
        if rank == 0:
            records = sorted((r for part in gathered for r in part), key=lambda t: t[0])
    out = [r for _, r in records]
    if rank == 0 and summary is not None:
        summary.update(summarise_avs(out))
    for rec in out:
        rec.pop("_avss", None)                                          # 3 x classes floats per sample: summed above, not part of the record
    if rank == 0:
        for rec in out:
            if on_result is not None:
                on_result(rec)
        if out_path is not None:
            with open(out_path, "a") as f:
                for rec in out:
                    f.write(json.dumps(rec) + "\n")
    return out
