"""Eval-harness counterpart (crab_amd/harness.py) against tests/golden/harness.npz, which was produced by
running the reference's UnifiedTestDataset + DataCollatorForUnifiedTestDataset here (tests/golden/make_golden.py harness)."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from crab_amd import harness
from tests.util import MM_SPECIAL, load_fixture, tiny_tokenizer

QUESTION = "How many instruments are playing?"
# (task, audio samples, video frames) of the fixture's video + audio samples, in the order their windows were recorded
AV_SAMPLES = [("avqa", 6000, 37), ("avqa", 6031, 5), ("ave", 1000, 100), ("avvp", 1017, 8)]


def _tok():
    tok = tiny_tokenizer()
    tok.add_tokens(MM_SPECIAL, special_tokens=True)
    return tok


def _ramp(n):
    return (np.arange(n, dtype=np.float32) + 1.0) / n


def test_instructions_match_reference():
    meta, _ = load_fixture("harness")
    fields = {"avqa": dict(question=QUESTION), "ref-avs": dict(exp="The Dog")}
    for task, want in meta["instructions"].items():
        for w in want:
            assert harness.build_instruction(task, **fields.get(task, {})) == w, task
    with pytest.raises(ValueError, match="invalid task"):
        harness.build_instruction("nope")
    with pytest.raises(ValueError):
        harness.build_instruction("avqa")


def test_prompts_ids_labels_match_reference():
    meta, A = load_fixture("harness")
    tok = _tok()
    col = harness.Collator(tok)
    for key, prompt, output in meta["prompts"]:
        task = key.rstrip("0123456789")
        ins, out = harness.wrap_prompt(tok, harness.build_instruction(task, question=QUESTION if task == "avqa" else None))
        assert ins == prompt and out == output
        b = col([{"instruction": ins, "output": out, "task_name": task, "video": torch.zeros(1), "audio": torch.zeros(1), "video_path": "v"}])
        assert torch.equal(b["batch_input_ids"][0], A[key + "_ids"].long())
        assert torch.equal(b["batch_labels"][0], A[key + "_labels"].long())
        assert b["batch_task_names"] == [task] and sorted(b["batch_X_modals"][0]) == ["<audio>", "<video>"]
        assert b["batch_metadata"][0] == {"instruction": ins, "output": out, "video_path": "v", "audio_path": ""}
    assert tok.batch_decode(meta["decode_ids"], skip_special_tokens=False) == meta["decoded"]
    # a tokenizer without a chat template leaves prompt and target untouched (quick_start_dataset.py:284)

    class Plain:
        pass
    assert harness.wrap_prompt(Plain(), "x", "y") == ("x", "y")


def test_frame_sampling_matches_reference():
    meta, _ = load_fixture("harness")
    for (_, _, vlen), want in zip(AV_SAMPLES, meta["frames"]):
        assert harness.frame_indices(vlen, 8) == want
    assert harness.frame_indices(3, 10) == [0, 1, 2]
    with pytest.raises(ValueError):
        harness.frame_indices(0, 8)


def test_audio_windows_match_reference():
    meta, A = load_fixture("harness")
    k = 0
    for task, n, _ in AV_SAMPLES:
        win = harness.audio_windows(task, _ramp(n))
        assert win.dtype == torch.float32 and win.shape[0] == 10
        for w in win:
            ref = A[f"win_{k}"]
            assert w.shape[0] == ref.shape[0] == meta["n_windows"][k]
            assert torch.equal(w, ref.float()), (task, k)
            k += 1
    assert k == len(meta["n_windows"])
    # single-window tasks: second idx of a 5 s (s4 / ms3 / arig) or 10 s (avss) clip; only arig pads a short tail
    a = _ramp(1003)
    assert torch.equal(harness.audio_windows("s4", a, idx=2)[0], torch.from_numpy(a[400:600]))
    assert torch.equal(harness.audio_windows("avss", a, idx=9)[0], torch.from_numpy(a[900:1000]))
    assert harness.audio_windows("arig", a[:950], idx=4).shape == (1, 190)
    short = harness.audio_windows("arig", torch.from_numpy(a[:7]), idx=4)          # nps = 1, slice past the end -> zero padded
    assert short.shape == (1, 1)
    with pytest.raises(ValueError):
        harness.audio_windows("s4", a)


class _FakeModel:
    """generate() returns the first three prompt ids of every sample and records its keyword arguments."""

    def __init__(self):
        self.calls = []

    def generate(self, batch_input_ids, batch_labels, batch_X_modals, batch_task_names, **kw):
        self.calls.append(dict(kw, n=len(batch_input_ids)))
        return torch.stack([i[:3] for i in batch_input_ids])

    def generate_batches(self, batches, **kw):
        self.calls.append(dict(kw, n=[len(b["batch_input_ids"]) for b in batches], many=True))
        return [torch.stack([i[:3] for i in b["batch_input_ids"]]) for b in batches]


def _batches(tok, n):
    col = harness.Collator(tok)
    out = []
    for i in range(n):
        ins, o = harness.wrap_prompt(tok, harness.build_instruction("avqa", question=QUESTION + " " * i))
        out.append(col([{"instruction": ins, "output": o, "task_name": "avqa", "video": torch.full((1,), float(i)), "audio": torch.zeros(1)}]))
    return out


def test_run_inference_single_rank(tmp_path):
    tok = _tok()
    model = _FakeModel()
    path = str(tmp_path / "infer_results.jsonl")
    seen = []
    recs = harness.run_inference(_batches(tok, 3), model, tok, max_new_tokens=7, out_path=path, device="cpu", on_result=seen.append)
    assert len(recs) == 3 and seen == recs
    assert all(c["use_cache"] is True and c["max_new_tokens"] == 7 and c["n"] == 1 for c in model.calls)
    lines = [json.loads(l) for l in open(path)]
    assert lines == recs and all(set(r) == {"instruction", "output", "video_path", "audio_path", "predict"} for r in recs)
    assert recs[0]["predict"] == tok.batch_decode([tok.convert_tokens_to_ids(tok.tokenize(recs[0]["instruction"]))[:3]])[0]


def test_run_inference_with_batches_in_flight_returns_the_same_records():
    tok = _tok()
    one, many = _FakeModel(), _FakeModel()
    recs1 = harness.run_inference(_batches(tok, 5), one, tok, max_new_tokens=7, device="cpu")
    recs3 = harness.run_inference(_batches(tok, 5), many, tok, max_new_tokens=7, device="cpu", in_flight=3)
    assert recs1 == recs3 and len(recs3) == 5
    assert [c.get("many", False) for c in many.calls] == [True, True] and many.calls[0]["n"] == [1, 1, 1] and many.calls[1]["n"] == [1, 1]
    assert all(c["max_new_tokens"] == 7 and c["use_cache"] is True for c in many.calls)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tok = _tok()
    model = _FakeModel()
    recs = harness.run_inference(_batches(tok, 5), model, tok, max_new_tokens=4, device="cpu", rank=rank, world=world)
    q.put((rank, len(model.calls), [r["instruction"] for r in recs]))
    dist.barrier()
    dist.destroy_process_group()


def test_run_inference_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, ncalls, ins = q.get(timeout=180)
        got[r] = (ncalls, ins)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    tok = _tok()
    want = [b["batch_metadata"][0]["instruction"] for b in _batches(tok, 5)]
    assert got[0] == (3, want)                                # batches 0, 2, 4 ran on rank 0, which holds all five records in order
    assert got[1] == (2, [want[1], want[3]])


def _worker3(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tok = _tok()
    model = _FakeModel()
    recs = harness.run_inference(_batches(tok, 7), model, tok, max_new_tokens=4, device="cpu", rank=rank, world=world, in_flight=2)
    q.put((rank, [c["n"] if c.get("many") else [1] for c in model.calls], [r["instruction"] for r in recs]))
    dist.barrier()
    dist.destroy_process_group()


def test_run_inference_uneven_shards_world3_gloo_in_flight():
    """7 eval batches over 3 ranks (3 + 2 + 2: the batch count is not a multiple of the world size), two batches in flight per rank:
    rank 0 ends up with all seven records in batch order, the other ranks with their own (scripts/finetune/inference_hyper_lora.py:1466-1479
    is the single-process loop this shards)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker3, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(3):
        r, calls, ins = q.get(timeout=180)
        got[r] = (calls, ins)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    tok = _tok()
    want = [b["batch_metadata"][0]["instruction"] for b in _batches(tok, 7)]
    assert got[0][1] == want and got[0][0] == [[1, 1], [1]]           # batches 0, 3 in flight together, then 6 alone
    assert got[1][1] == [want[1], want[4]] and got[2][1] == [want[2], want[5]]


@pytest.mark.gpu
def test_make_instance_pipeline_gpu():
    """uint8 frames + waveform -> make_instance (device front-end) -> Collator -> run_inference on the tiny fixture model: the
    video tensor matches what the reference's dataset produced for the same constant frames, and the harness returns
    exactly what a direct generate() call returns."""
    from tests.util import weights_from_table
    hm, A = load_fixture("harness")
    meta, _ = load_fixture("full_tiny_llama")
    tok = tiny_tokenizer(pad_to=meta["base_vocab"])
    assert len(tok) == meta["base_vocab"]
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    from tests.util import bert_cfg
    cfg = UnifiedConfig(**meta["dec"], pad_token_id=meta["pad_token_id"])
    cfg.vocab_size = meta["base_vocab"]
    model = get_peft_model(UnifiedForCausalLM(cfg, device="cuda"), LoraConfig())
    model.get_model().pad_token_id = meta["pad_token_id"]
    model.get_model().init_multimodal_modules(d_model=meta["d_model"], visual_branch=True, audio_branch=True, select_layer_list=meta["select"],
                                              clip_config=meta["clip"], beats_config=meta["beats"], bert_config=bert_cfg(meta["qf"]))
    model.initialize_MM_tokenizer(tok, mask_token_nums=6)      # the real tokenizer object: ids come from tokenizer.add_tokens
    assert model.SPECIAL_TOKEN_2_IDS == meta["special"]
    assert tok.convert_tokens_to_ids("<video>") == meta["special"]["<video>"]
    r = model.load_state_dict(weights_from_table(meta), strict=False)
    assert not r.missing_keys
    insts = []
    for i, (task, n, vlen) in enumerate(AV_SAMPLES[:2]):
        frames = [np.full((224, 224, 3), j % 256, np.uint8) for j in range(vlen)]
        inst = harness.make_instance(task, tok, question=QUESTION, frames=frames, audio=_ramp(n * 16), n_frames=8)
        assert inst["video"].shape == (min(8, vlen), 3, 224, 224) and inst["audio"].shape[0] == 10 and inst["audio"].shape[2] == 128
        np.testing.assert_allclose(inst["video"].mean(dim=(1, 2, 3)).cpu().numpy(), A[f"avqa{i}_video_mean"].numpy(), atol=2e-6)
        insts.append(inst)
    col = harness.Collator(tok)
    batches = [col([insts[0]]), col([insts[1]])]
    recs = harness.run_inference(batches, model, tok, max_new_tokens=6, do_sample=False, pad_token_id=2, eos_token_id=None)
    assert len(recs) == 2
    for b, rec in zip(batches, recs):
        s = harness.to_device({k: v for k, v in b.items() if k != "batch_metadata"})
        ids = model.generate(**s, use_cache=True, max_new_tokens=6, do_sample=False, pad_token_id=2, eos_token_id=None)
        assert ids.shape == (1, 6)
        assert rec["predict"] == tok.batch_decode(ids, skip_special_tokens=False)[0]
    # both batches in flight together (through the Peft wrapper -> UnifiedForCausalLM.generate_batches): the same records
    recs2 = harness.run_inference(batches, model, tok, max_new_tokens=6, do_sample=False, pad_token_id=2, eos_token_id=None, in_flight=2)
    assert recs2 == recs
    # image task: .resize((224, 224)) + processor, one audio window
    img = np.random.RandomState(0).randint(0, 256, (120, 300, 3), dtype=np.uint8)
    it = harness.make_instance("s4", tok, image=img, audio=_ramp(16000 * 5), idx=3, mask=torch.zeros(1, 224, 224))
    assert it["image"].shape == (1, 3, 224, 224) and it["audio"].dim() == 2 and it["audio"].shape[1] == 128 and "mask" in it
    from PIL import Image
    from crab_amd.frontend import CLIPImageProcessor
    ref = np.asarray(Image.fromarray(img).resize((224, 224)))
    assert np.array_equal(CLIPImageProcessor().resize_exact(img, 224, 224).cpu().numpy(), ref)          # bit-exact with Pillow


def test_summarise_avs_is_the_closing_arithmetic_of_the_pixel_task_loops():
    """scripts/quick_start.py:120-135 (miou: an fp32 running sum / count; f-score: Python floats), :343-358 (ms), :437-447 (avss: per-class
    sums / counts, NaN -> 0, class means with and without the last class) over the records run_inference_avs produces."""
    from oracle import metrics_oracle as MO
    A = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seg_metrics.npz")))
    recs = [{"iou": float(v), "fscore": float(f)} for v, f in zip(A["iou_each"], A["f_each"])] + [{"pred_path": None}]
    recs += [{"s": float(v)} for v in A["s_each"]]
    per_frame = [MO.batch_miou_fscore(A["cls_pred"][f:f + 1], A["cls_tgt"][f:f + 1])[:3] for f in range(A["cls_pred"].shape[0])]
    recs += [{"_avss": [v.tolist() for v in t]} for t in per_frame]
    out = harness.summarise_avs(recs)
    acc = np.float32(0)
    for v in A["iou_each"]:
        acc = np.float32(acc + v)
    assert out["count"] == 4 and out["miou"] == float(acc / np.float32(4)) and out["f_score"] == float(sum(A["f_each"].tolist()) / 4)
    assert out["count_null"] == 4 and out["ms"] == sum(float(v) for v in A["s_each"]) / 4
    want = MO.avss_final(A["cls_miou"], A["cls_fscore"], A["cls_count"])          # the reference's batch call sums the frames the same way
    assert out["avss"]["count"] == 3 and all(abs(out["avss"][k] - want[k]) < 1e-7 for k in want)
    assert harness.summarise_avs([{"pred_path": None}]) == {}
