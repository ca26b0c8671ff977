"""Stateful fuzz at the MODEL level (r06, next to fuzz_engine_state.py): a random sequence of UnifiedForCausalLM calls - prepare_multimodal_inputs, generate,
generate_batches in flight and coalesced, the pixel loops' generate_avs_many halves (ragged decode with hidden states + the batched SegModule) - with changing batch sizes, frame counts, audio windows (1-s and 2-s), modality subsets and ragged prompts on ONE
tiny Crab (CLIP tower + BEATs + both Q-Former projectors + hyper-LoRA decoder), whose encoders, projectors and engine keep workspaces, tables and graphs between
calls.  Two checkpoints alternate: the carried model reloads (load_state_dict) whenever a call names the other one.  Three executions of the same sequence must
agree BIT FOR BIT: a freshly built model per call, one model straight through, and one model straight through on a side HIP stream.   python scripts/fuzz_model_state.py [calls] [seed]"""
import os, random, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import synth

BF = torch.bfloat16
warnings.filterwarnings("ignore", category=RuntimeWarning)
_argv = sys.argv[1:] if __name__ == "__main__" else []
NCALL = int(_argv[0]) if len(_argv) > 0 else 40
SEED = int(_argv[1]) if len(_argv) > 1 else 0
rng = random.Random(SEED)
META = dict(
    dec=dict(hidden_size=128, intermediate_size=136, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, vocab_size=320, rms_norm_eps=1e-5, rope_theta=10000.0),
    clip=dict(hidden_size=128, intermediate_size=256, num_hidden_layers=4, num_attention_heads=2, image_size=224, patch_size=14, layer_norm_eps=1e-5),
    select=[2, 3, 4],
    beats=dict(input_patch_size=16, embed_dim=64, encoder_embed_dim=128, encoder_ffn_embed_dim=136, encoder_attention_heads=2, encoder_layers=2, conv_pos=128,
               conv_pos_groups=16, num_buckets=320, max_distance=800, deep_norm=True, gru_rel_pos=True, conv_bias=False, relative_position_embedding=True,
               layer_norm_first=False, activation_fn="gelu", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0,
               finetuned_model=False),
    qf=dict(hidden=128, heads=2, inter=136), d_model=128, base_vocab=303, pad_token_id=2, qkv_bias=False)


def build():
    """tests/util.build_tiny_crab + the SegModule (scripts/quick_start.py:505-529 with segment_branch)"""
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    from tests.util import DuckTokenizer, bert_cfg
    cfg = UnifiedConfig(**META["dec"], pad_token_id=META["pad_token_id"])
    cfg.vocab_size = META["base_vocab"]
    model = get_peft_model(UnifiedForCausalLM(cfg, device="cuda"), LoraConfig())
    model.get_model().pad_token_id = META["pad_token_id"]
    model.get_model().init_multimodal_modules(d_model=META["d_model"], visual_branch=True, audio_branch=True, segment_branch=True, select_layer_list=META["select"],
                                              clip_config=META["clip"], beats_config=META["beats"], bert_config=bert_cfg(META["qf"]),
                                              vit_image_embedding_dim=META["clip"]["hidden_size"])
    model.initialize_MM_tokenizer(DuckTokenizer(META["base_vocab"]), mask_token_nums=6)
    return model


def weights(seed=900):
    torch.manual_seed(seed)
    model = build()
    W = {}
    for k, v in model.state_dict().items():
        if not v.dtype.is_floating_point:
            W[k] = v.clone().cpu(); continue
        if v.dim() > 1:
            t = torch.randn(v.shape) * min(0.2, 1.2 / v[0].numel() ** 0.5)
        elif k.endswith("weight") and ("norm" in k.lower() or "ln" in k.lower().split(".")[-2] or "layer_norm" in k.lower()):
            t = 1 + 0.1 * torch.randn(v.shape)
        else:
            t = 0.05 * torch.randn(v.shape)
        if k.endswith("weight_g"): t = t.abs() + 0.5
        W[k] = t.to(BF).float()
    for k in list(W):                                      # BEATs: every layer aliases layer 0's relative-position table (backbone.py:78-81)
        if k.endswith("self_attn.relative_attention_bias.weight") and ".layers.0." not in k:
            W[k] = W[k.split(".layers.")[0] + ".layers.0.self_attn.relative_attention_bias.weight"]
    return W


def new_model(W):
    model = build()
    r = model.load_state_dict(W, strict=False)
    assert not r.missing_keys, r.missing_keys[:4]
    return model


def make_calls(n):
    calls = []
    for i in range(n):
        kind = rng.choice(["prepare", "generate", "generate", "batches", "avs"])
        c = dict(kind=kind, seed=i, tv=rng.choice([1, 2, 4, 8]), ta=rng.choice([1, 3, 10]), la=rng.choice([98, 198]), mods=rng.choice(["va", "va", "va", "v", "a", "none", "image"]),
                 n=rng.choice([1, 3, 5]), wset=int(i >= n // 3) ^ int(rng.random() < 0.15))      # the checkpoint in force: mostly the first, then mostly the second
        if kind == "avs":
            c.update(N=rng.choice([1, 2, 3, 5]), max_rows=rng.choice([None, None, 2]), mods="image_a", n=8)
        elif kind == "batches":
            c.update(sizes=[rng.choice([1, 2, 3]) for _ in range(rng.choice([1, 2, 4]))], coalesce=rng.random() < 0.6, max_rows=rng.choice([None, None, 3]))
        else:
            c.update(B=rng.choice([1, 2, 3, 5]))
        calls.append(c)
    return calls


def sample(um, c, clip):
    sp = um.SPECIAL_TOKEN_2_IDS
    ids = synth.synth_prompt_ids(12 + (clip * 5) % 9, 303, sp, seed=c["seed"], clip=clip)
    drop = []
    if c["mods"] in ("a", "none"): drop += ["<video_start>", "<video>", "<video_end>"]
    if c["mods"] in ("v", "none", "image"): drop += ["<audio_start>", "<audio>", "<audio_end>"]
    for t in drop:
        ids = ids[ids != sp[t]]
    mods = {}
    if c["mods"] in ("image", "image_a"):
        for a_, b_ in (("<video_start>", "<image_start>"), ("<video>", "<image>"), ("<video_end>", "<image_end>")):
            ids[ids == sp[a_]] = sp[b_]
        mods['<image>'] = synth.synth_video(1, seed=c["seed"], clip=clip)
    if c["mods"] in ("va", "v"): mods['<video>'] = synth.synth_video(c["tv"], seed=c["seed"], clip=clip)
    if c["mods"] in ("va", "a", "image_a"): mods['<audio>'] = synth.synth_audio(c["ta"], c["la"], seed=c["seed"], clip=clip)
    return ids, mods


def run(model, c):
    um = model.base_model.model
    kw = dict(max_new_tokens=c["n"], min_new_tokens=c["n"], pad_token_id=2, eos_token_id=None, use_cache=True)
    if c["kind"] == "avs":                                  # the pixel loops: N one-sample calls as one ragged batch, masks from the last six steps (bench.py's pick rule)
        smp = []
        for j in range(c["N"]):
            ids, mods = sample(um, c, j)
            smp.append(dict(batch_input_ids=[ids], batch_labels=[torch.full_like(ids, -100)], batch_X_modals=[mods],
                            batch_task_names=['avss' if (j + c["seed"]) % 3 == 0 else ('s4', 'ms3', 'ref-avs')[j % 3]]))
        inputs, outs = um._avs_generate(smp, c["max_rows"], kw)
        res = um._avs_segment(smp, inputs, outs, [(g_, 0, list(range(c["n"] - 7, c["n"] - 1))) for g_ in range(c["N"])])
        out = []
        for r_ in res:
            out += [r_['output_ids'].clone(), r_['pred_masks'][0].clone()]
        return out
    if c["kind"] == "batches":
        batches, clip = [], 0
        for b in c["sizes"]:
            xs = [sample(um, c, clip + j) for j in range(b)]
            clip += b
            batches.append(dict(batch_input_ids=[x[0] for x in xs], batch_labels=[torch.full_like(x[0], -100) for x in xs], batch_X_modals=[x[1] for x in xs],
                                batch_task_names=['avqa'] * b))
        r = model.generate_batches(batches, coalesce=c["coalesce"], max_rows=c["max_rows"] if c["coalesce"] else None, **kw)
        return [x.clone() for x in r]
    xs = [sample(um, c, j) for j in range(c["B"])]
    args = dict(batch_input_ids=[x[0] for x in xs], batch_labels=[torch.full_like(x[0], -100) for x in xs], batch_X_modals=[x[1] for x in xs],
                batch_task_names=['avqa'] * c["B"])
    if c["kind"] == "prepare":
        d = um.prepare_multimodal_inputs(**args)
        return [d["inputs_embeds"].clone(), d["attention_mask"].clone(), d["position_ids"].clone()]
    r = model.generate(**args, output_logits=True, return_dict_in_generate=True, **kw)
    return [r.sequences.clone(), torch.stack(r.logits, 1).clone()]


def same(a, b):
    return len(a) == len(b) and all(x.shape == y.shape and torch.equal(x, y) for x, y in zip(a, b))


if __name__ == "__main__":
    bad = []
    WS = [weights(900), weights(901)]                      # two checkpoints: the carried model RELOADS (load_state_dict) when a call names the other one -
    calls = make_calls(NCALL)                              # packed q|k|v / [W | B] / norm copies, tables and graphs made from the old weights must not survive
    fresh = []
    for c in calls:
        m = new_model(WS[c["wset"]])
        try:
            fresh.append(run(m, c))
        except Exception as e:      # noqa: BLE001
            fresh.append(e)
        del m
    for label, stream in (("carried", None), ("carried, side stream", torch.cuda.Stream())):
        m, cur, reloads = new_model(WS[0]), 0, 0
        n_same = 0
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for i, c in enumerate(calls):
                desc = f"{label}: call {i}: {c}"
                try:
                    if c["wset"] != cur:
                        r_ = m.load_state_dict(WS[c["wset"]], strict=False)
                        assert not r_.missing_keys
                        cur, reloads = c["wset"], reloads + 1
                    got = run(m, c)
                except Exception as e:      # noqa: BLE001
                    if not isinstance(fresh[i], Exception) or type(fresh[i]) is not type(e):
                        bad.append(desc + f" -> {type(e).__name__}: {str(e)[:200]} (fresh model: {'ok' if not isinstance(fresh[i], Exception) else type(fresh[i]).__name__})")
                    else:
                        n_same += 1
                    continue
                if isinstance(fresh[i], Exception):
                    bad.append(desc + f" -> ok here, {type(fresh[i]).__name__}: {str(fresh[i])[:160]} on a fresh model"); continue
                torch.cuda.synchronize()
                if same(got, fresh[i]): n_same += 1
                else: bad.append(desc + f" -> results {[k for k, (a, b) in enumerate(zip(got, fresh[i])) if a.shape != b.shape or not torch.equal(a, b)]} differ from the fresh model's")
        torch.cuda.synchronize()
        print(f"{label}: {n_same} of {len(calls)} calls bit-identical to a freshly built model ({reloads} checkpoint reloads on the way; {sum(isinstance(f, Exception) for f in fresh)} calls raise on both)", flush=True)
        del m
    print(f"{3 * NCALL} calls computed, {len(bad)} failures")
    for b_ in bad[:40]: print("FAIL", b_)
    sys.exit(1 if bad else 0)
