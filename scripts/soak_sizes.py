"""r06: the full-size model at sizes nothing else visits - batch sizes off every tile / regime boundary, batches beyond one decode group, long prompts,
the reference's max_new_tokens = 500, ragged waves, pixel batches of odd sizes, the Qwen2 decoder at an odd batch.  (The 512-sample pixel call of
bench.py found a workspace query that was not monotone; this script looks for its relatives.)  Checks per case: no error, the result's shape, ids inside
the vocabulary, and - where the same clips are decoded in two ways - that the first clips' ids agree up to sub-margin flips (a flip needs a top-2 margin
below 2 % of the logit scale in the reference call's own logits).   python scripts/soak_sizes.py [llama|qwen|avs|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import synth
from crab_amd.build_model import build_crab

what = sys.argv[1] if len(sys.argv) > 1 else "all"
fails = []


def inputs(model, B, frames=8, nt=128, l_a=98, t_a=10, clip0=0, ragged=False):
    tab = model.base_model.model.SPECIAL_TOKEN_2_IDS
    ids = [synth.synth_prompt_ids(nt + ((i * 7) % 23 if ragged else 0), model.base_vocab, tab, clip=clip0 + i).cuda() for i in range(B)]
    mods = [{'<video>': synth.synth_video(frames, clip=clip0 + i).cuda(), '<audio>': synth.synth_audio(t_a, l_a, clip=clip0 + i).cuda()} for i in range(B)]
    return dict(batch_input_ids=ids, batch_labels=[torch.full_like(i, -100) for i in ids], batch_X_modals=mods, batch_task_names=['avqa'] * B)


def case(name, fn):
    t0 = time.perf_counter()
    try:
        msg = fn()
        torch.cuda.synchronize()
        print(f"ok   {name}: {msg}  ({time.perf_counter() - t0:.1f} s, {torch.cuda.max_memory_allocated() / 2 ** 30:.0f} GiB peak)", flush=True)
    except Exception as e:      # noqa: BLE001
        fails.append(name)
        print(f"FAIL {name}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    torch.cuda.reset_peak_memory_stats()


def agree(a, b, ref_logits, n):
    """rows of a and b (the same clips decoded in two batches, b with its per-step logits): equal, or first difference at a step whose top-2 margin in b's own
    logits is below 2 % of their scale.  (Both calls must see the same inputs_embeds: the encoders' kernels differ between a 1-clip and a >= 2-clip call -
    4e-3 of the embedding scale, which a random-weight 7B decoder amplifies to 2-3 % of the logit scale, scripts/exp/batch_noise_debug.py - so the reference
    call is a batch too.)"""
    bad = 0
    for i in range(n):
        if torch.equal(a[i], b[i]):
            continue
        j = int((a[i] != b[i]).nonzero()[0])
        lg = ref_logits[j][i].float()
        top2 = lg.topk(2).values
        if float(top2[0] - top2[1]) > 0.02 * float(lg.abs().max()):
            bad += 1
    return bad


def decoder_cases(llm):
    model = build_crab(llm)
    um = model.base_model.model
    V = um.lm_head.weight.shape[0]
    kw = dict(use_cache=True, eos_token_id=None, pad_token_id=um.model.pad_token_id)

    def gen(B, new, **ikw):
        x = inputs(model, B, **ikw)
        r = model.generate(**x, max_new_tokens=new, min_new_tokens=new, **kw)
        assert tuple(r.shape) == (B, new) and int(r.min()) >= 0 and int(r.max()) < V
        return x, r

    def odd_batches():
        out = []
        x8 = inputs(model, 8)
        r8 = model.generate(**x8, max_new_tokens=6, min_new_tokens=6, output_logits=True, return_dict_in_generate=True, **kw)
        for B in (3, 17, 65, 129, 257, 300, 511):
            x, r = gen(B, 6)
            n = min(8, B)
            bad = agree(r[:n], r8.sequences[:n], r8.logits, n)
            assert bad == 0, f"B = {B}: {bad} rows differ from the 8-clip call beyond a sub-margin flip"
            out.append(B)
        return f"batches {out} agree with the 8-clip call on their first rows"
    case(f"{llm}: odd batch sizes", odd_batches)
    case(f"{llm}: 700 clips in one call (two decode groups)", lambda: f"groups {gen(700, 4) and um._engine.last_plan['groups']}")
    case(f"{llm}: 37 clips x 500 new tokens (the reference's max_new_tokens)", lambda: f"shape {tuple(gen(37, 500)[1].shape)}")
    case(f"{llm}: long prompt (24 frames, 900 text tokens, 2-s audio windows: S ~ 2300), 21 clips", lambda: f"shape {tuple(gen(21, 5, frames=24, nt=900, l_a=198)[1].shape)}")
    case(f"{llm}: one frame, one audio window, 5-token prompt, 1 new token", lambda: f"shape {tuple(gen(2, 1, frames=1, nt=8, t_a=1)[1].shape)}")

    def ragged():
        out = []
        for n_clips, rows in ((41, 128), (130, 64), (600, 512)):
            x = inputs(model, n_clips, ragged=True, frames=4)
            calls = [dict(batch_input_ids=x["batch_input_ids"][i:i + 1], batch_labels=x["batch_labels"][i:i + 1], batch_X_modals=x["batch_X_modals"][i:i + 1],
                          batch_task_names=['avqa']) for i in range(n_clips)]
            res = model.generate_batches(calls, coalesce=True, max_rows=rows, max_new_tokens=5, min_new_tokens=5, **kw)
            assert len(res) == n_clips and all(tuple(r.shape) == (1, 5) for r in res)
            ref = model.generate_batches(calls[:6], coalesce=True, max_new_tokens=5, min_new_tokens=5, **kw)          # the same six clips as a wave of their own
            same = sum(int(torch.equal(a_, b_)) for a_, b_ in zip(res[:6], ref))
            assert same >= 4, f"{n_clips} ragged clips in waves of {rows}: only {same} of the first 6 rows equal their 6-row wave (sub-margin flips are 0-1 per 6 rows x 5 steps)"
            out.append((n_clips, rows, um._engine.last_plan.get("groups")))
        return f"(clips, rows per wave, plan) {out}"
    case(f"{llm}: ragged coalesced waves", ragged)
    del model, um
    torch.cuda.empty_cache()


def avs_cases():
    model = build_crab("llama", segment=True)
    um = model.base_model.model
    inner = model.get_model()
    g = torch.Generator(device="cuda").manual_seed(77)
    for name, buf in inner.seg_module.named_buffers():
        if name.endswith("positional_encoding_gaussian_matrix"):
            buf.normal_(generator=g)
    sp = um.SPECIAL_TOKEN_2_IDS
    NEW = 12

    def samples(N):
        out = []
        for i in range(N):
            ids = synth.synth_prompt_ids(48 + (i % 5), model.base_vocab, sp, clip=7000 + i)
            for a_, b_ in (("<video_start>", "<image_start>"), ("<video>", "<image>"), ("<video_end>", "<image_end>")):
                ids[ids == sp[a_]] = sp[b_]
            task = 'avss' if i % 4 == 3 else ('s4', 'ms3', 'ref-avs')[i % 3]
            out.append({"batch_input_ids": [ids.cuda()], "batch_labels": [torch.full_like(ids, -100)],
                        "batch_X_modals": [{'<image>': synth.synth_video(1, clip=7000 + i).cuda(), '<audio>': synth.synth_audio(1, 98, clip=7000 + i).cuda()}],
                        "batch_task_names": [task]})
        return out
    kw = dict(max_new_tokens=NEW, min_new_tokens=NEW, pad_token_id=um.model.pad_token_id, eos_token_id=um.config.eos_token_id, use_cache=True)

    def pixel(N, max_rows=None):
        s = samples(N)
        chosen = [(g_, 0, list(range(NEW - 7, NEW - 1))) for g_ in range(N)]          # a random decoder never emits <mask_i>: the last six steps stand in (bench.py)
        inputs_, outs = um._avs_generate(s, max_rows, kw)
        res = um._avs_segment(s, inputs_, outs, chosen)
        assert len(res) == N
        for r, sm in zip(res, s):
            m = r['pred_masks'][0]
            assert tuple(m.shape) == ((71, 224, 224) if sm["batch_task_names"][0] == 'avss' else (1, 224, 224)) and bool(torch.isfinite(m.float()).all())
        return f"{N} samples, plan {um._engine.last_plan.get('groups')}"
    for N, mr in ((1, None), (7, None), (129, None), (300, None), (513, None), (70, 32)):
        case(f"avs: {N} samples per call" + (f", waves of {mr}" if mr else ""), lambda N=N, mr=mr: pixel(N, mr))


if what in ("llama", "all"): decoder_cases("llama")
if what in ("qwen", "all"): decoder_cases("qwen")
if what in ("avs", "all"): avs_cases()
print(f"{len(fails)} failures" + ("" if not fails else ": " + "; ".join(fails)))
sys.exit(1 if fails else 0)
