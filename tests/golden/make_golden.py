#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/*.npz by running the REFERENCE itself.

Runs only in the build container (needs /root/reference; read-only, imported through
tests/golden/ref_shims.py, nothing copied).  For every hot-path component it
  1. builds the reference module at a tiny configuration,
  2. loads crab_amd.synth weights (seeded; regenerated, never stored),
  3. runs the reference forward in fp32 eager on CPU,
  4. stores inputs' seeds, the (name, shape, checksum) weight table and the reference outputs.
tests/test_oracle_golden.py then pins oracle/crab_oracle.py against these files, and the GPU parity
tests compare the HIP path with the same numbers.

    python tests/golden/make_golden.py            # rewrite ALL fixtures (every entry of tests/golden/registry.py, one interpreter each)
    python tests/golden/make_golden.py fullwidth  # the six full-width fixtures (r06)
    python tests/golden/make_golden.py metrics    # one generator, in this process
"""
from __future__ import annotations

import sys

sys.dont_write_bytecode = True          # before ANY import that could touch /root/reference: the reference tree stays untouched

import importlib.machinery
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import ref_shims  # noqa: E402
from crab_amd import synth  # noqa: E402

SEED = 1234
torch.set_grad_enabled(False)
torch.manual_seed(0)


def canon(name: str) -> str:
    """Canonical key = the name under the reference's pinned transformers==4.37.2.  transformers 5.x
    flattened CLIPVisionModel (no `.vision_model.` level); weights are named the 4.37.2 way."""
    if "vision_tower." in name and "vision_tower.vision_model." not in name:
        name = name.replace("vision_tower.", "vision_tower.vision_model.")
    return name


def load_synth(module: torch.nn.Module, prefix: str, seed: int = SEED, alias_groups=()):
    """Fill every state_dict entry of `module` with synth weights named canon(prefix + key).
    Returns the (name, shape, checksum) table."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        if not v.dtype.is_floating_point:
            continue
        new[k] = synth.synth_tensor(canon(prefix + k), v.shape, seed)
    for src_key, others in alias_groups:           # tied parameters (BEATs relative_attention_bias)
        for o in others:
            if o in new:
                new[o] = new[src_key].clone()
    missing = module.load_state_dict(new, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    sd2 = module.state_dict()
    table = []
    for k, v in sd2.items():
        if v.dtype.is_floating_point and "_Qformer.cls." not in k:   # LM head: unused, tied params
            table.append((canon(prefix + k), list(v.shape), synth.checksum(v)))
    return table


def save(name: str, meta: dict, **arrays):
    path = os.path.join(HERE, name + ".npz")
    arrs = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------ tiny configurations
TINY_DEC = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=2, vocab_size=320, rms_norm_eps=1e-5, rope_theta=10000.0)
TINY_QWEN = dict(hidden_size=256, intermediate_size=384, num_hidden_layers=2, num_attention_heads=4,
                 num_key_value_heads=2, vocab_size=320, rms_norm_eps=1e-6, rope_theta=1000000.0)
TINY_CLIP = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=6, num_attention_heads=2,
                 image_size=224, patch_size=14, layer_norm_eps=1e-5)
TINY_CLIP_SELECT = [2, 4, 5]
TINY_BEATS = dict(input_patch_size=16, embed_dim=64, encoder_embed_dim=128, encoder_ffn_embed_dim=256,
                  encoder_attention_heads=2, encoder_layers=3, conv_pos=128, conv_pos_groups=16,
                  num_buckets=320, max_distance=800, deep_norm=True, gru_rel_pos=True, conv_bias=False,
                  relative_position_embedding=True, layer_norm_first=False, activation_fn="gelu",
                  dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0,
                  dropout_input=0.0, finetuned_model=False)
TINY_QF = dict(hidden=128, heads=2, inter=256)
D_MODEL = 128


def build_lora_linear():
    from peft_hyper.tuners.lora import Linear
    lin = Linear(128, 256, r=8, lora_alpha=16, lora_nums=3, lora_dropout=0.05, bias=True)
    lin.eval()
    table = load_synth(lin, "lin.")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 5, 128, generator=g)
    y = lin(x)
    save("hyperlora_linear", dict(seed=SEED, in_features=128, out_features=256, r=8, lora_alpha=16, lora_nums=3,
                                  table=table, x_seed=5), x=x, y=y)


def build_beats(cfg_dict):
    from models.beats.BEATs import BEATs, BEATsConfig
    m = BEATs(BEATsConfig(cfg_dict)).eval()
    m.training = False
    return m


def beats_alias(m):
    L = len(m.encoder.layers)
    return [("encoder.layers.0.self_attn.relative_attention_bias.weight",
             [f"encoder.layers.{i}.self_attn.relative_attention_bias.weight" for i in range(1, L)])]


def golden_beats():
    m = build_beats(TINY_BEATS)
    table = load_synth(m, "model.audio_encoder.audio_encoder.", alias_groups=beats_alias(m))
    outs = {}
    for L in (98, 198):
        x = synth.synth_audio(3, L, seed=SEED, clip=L)
        pm = torch.zeros(x.shape[:-1]).bool()
        y, _ = m.extract_features(x, padding_mask=pm, feature_only=True)
        outs[f"x{L}"] = x
        outs[f"y{L}"] = y
    save("beats_tiny", dict(seed=SEED, cfg=TINY_BEATS, table=table), **outs)
    # integer-exact bucket tables at the full configuration
    from models.beats.backbone import MultiheadAttention
    mha = MultiheadAttention(768, 12, self_attention=True, has_relative_attention_bias=True, num_buckets=320,
                             max_distance=800, gru_rel_pos=True)
    b = {}
    for n in (48, 96):
        ctx = torch.arange(n)[:, None]
        mem = torch.arange(n)[None, :]
        b[f"b{n}"] = mha._relative_positions_bucket(mem - ctx, bidirectional=True).to(torch.int32)
    save("beats_buckets", dict(num_buckets=320, max_distance=800), **b)


def build_clip(clip_cfg=None):
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg = CLIPVisionConfig(**(clip_cfg or TINY_CLIP), hidden_act="quick_gelu", attention_dropout=0.0, projection_dim=64)
    cfg._attn_implementation = "eager"
    return CLIPVisionModel(cfg).eval()


def build_visual_encoder(me, clip_cfg=None, select=None):
    class VE(me.VisualEncoder):
        def __init__(self, tower, select):
            torch.nn.Module.__init__(self)
            self.select_layer_list = select
            self.select_feature = 'patch'
            self.vision_tower = tower
    return VE(build_clip(clip_cfg), select or TINY_CLIP_SELECT).eval()


def golden_clip(me):
    ve = build_visual_encoder(me)
    table = load_synth(ve, "model.visual_encoder.")
    video = synth.synth_video(2, seed=SEED, clip=7)[None]            # [1,2,3,224,224]
    feats = ve(video)
    save("clip_tiny", dict(seed=SEED, cfg=TINY_CLIP, select=TINY_CLIP_SELECT, table=table, t_v=2, clip=7),
         f0=feats[0], f1=feats[1], f2=feats[2])


def golden_projectors(me):
    vl = me.VLProjector(hidden_size=128, image_token_nums=256, num_query_token=32, num_hidden_layers=2,
                        d_model=D_MODEL, depth=2).eval()
    tv = load_synth(vl, "model.vl_projector.")
    g = torch.Generator().manual_seed(11)
    feat = torch.randn(1, 2 * 256, 128, generator=g)
    yv = vl(feat)
    al = me.ALProjector(hidden_size=128, num_query_token=32, num_hidden_layers=2, d_model=D_MODEL, depth=2).eval()
    ta = load_synth(al, "model.al_projector.")
    g = torch.Generator().manual_seed(12)
    af = torch.randn(1, 3, 48, 128, generator=g)
    ya = al(af)
    save("projectors_tiny", dict(seed=SEED, qf=TINY_QF, d_model=D_MODEL, table=tv + ta, vseed=11, aseed=12),
         vfeat=feat, vout=yv, afeat=af, aout=ya)


class _Tok:
    """Duck-typed tokenizer for initialize_MM_tokenizer (unified_arch.py:409-459)."""
    def __init__(self, n):
        self.n = n
        self.added = []

    def __len__(self):
        return self.n

    def add_tokens(self, toks, special_tokens=False):
        self.added += list(toks)
        self.n += len(toks)
        return len(toks)


def build_full_model(me, dec_cfg, qwen=False):
    """tiny UnifiedForCausalLM -> get_peft_model -> encoders attached by hand -> initialize_MM_tokenizer."""
    from peft_hyper import LoraConfig, get_peft_model
    if qwen:
        from transformers import Qwen2Config, Qwen2ForCausalLM
        cfg = Qwen2Config(**dec_cfg, max_position_embeddings=2048, tie_word_embeddings=False,
                          use_sliding_window=False, attention_dropout=0.0)
        cfg._attn_implementation = "eager"
        base = Qwen2ForCausalLM(cfg)
    else:
        from models.unified_llama import UnifiedForCausalLM
        from transformers import LlamaConfig
        cfg = LlamaConfig(**dec_cfg, max_position_embeddings=2048, tie_word_embeddings=False, attention_bias=False,
                          pretraining_tp=1)
        cfg._attn_implementation = "eager"
        base = UnifiedForCausalLM(cfg)
    peft_config = LoraConfig(task_type="CAUSAL_LM",
                             target_modules="q_proj,k_proj,v_proj,o_proj,gate_proj,down_proj,up_proj".split(','),
                             inference_mode=False, r=8, lora_alpha=16, lora_dropout=0.05, lora_nums=3)
    model = get_peft_model(base, peft_config)
    return model, cfg


def golden_full(me):
    model, cfg = build_full_model(me, TINY_DEC)
    inner = model.get_model()
    inner.pad_token_id = 2
    inner.visual_encoder = build_visual_encoder(me)
    inner.vl_projector = me.VLProjector(hidden_size=128, image_token_nums=256, num_query_token=32,
                                        num_hidden_layers=2, d_model=D_MODEL, depth=2)
    class AE(me.AudioEncoder):
        def __init__(self, beats):
            torch.nn.Module.__init__(self)
            self.audio_encoder = beats
    beats = build_beats(TINY_BEATS)
    inner.audio_encoder = AE(beats)
    inner.al_projector = me.ALProjector(hidden_size=128, num_query_token=32, num_hidden_layers=2, d_model=D_MODEL,
                                        depth=2)
    tok = _Tok(TINY_DEC["vocab_size"] - 17)
    base_vocab = len(tok)
    model.base_model.model.initialize_MM_tokenizer(tok, mask_token_nums=6, use_vqgan=False)
    model.eval()
    alias = [("base_model.model.model.audio_encoder.audio_encoder." + c,
              ["base_model.model.model.audio_encoder.audio_encoder." + o for o in os_])
             for c, os_ in beats_alias(beats)]
    table = load_synth(model, "", alias_groups=alias)
    um = model.base_model.model
    tab = dict(um.SPECIAL_TOKEN_2_IDS)

    # ---- two samples of different length -> exercises left padding (appendix A.1 behaviour).
    # Greedy ids are only a meaningful bf16-vs-fp32 pin when top-2 margins exceed bf16 noise
    # (SURVEY.md 7 "hard parts"), so the prompt/clip indices are searched for comfortable margins.
    NEW = 12
    gen_kw = dict(use_cache=True, max_new_tokens=NEW, do_sample=False, output_logits=True,
                  return_dict_in_generate=True, pad_token_id=2, eos_token_id=None)
    from transformers import GenerationMixin  # noqa: F401

    def run(c0, c1):
        ids0 = synth.synth_prompt_ids(24, base_vocab, tab, seed=SEED, clip=c0)
        ids1 = synth.synth_prompt_ids(17, base_vocab, tab, seed=SEED, clip=c1)
        mods = []
        for c in (c0, c1):
            mods.append({'<video>': synth.synth_video(2, seed=SEED, clip=c),
                         '<audio>': synth.synth_audio(3, 98, seed=SEED, clip=c)})
        lab = [torch.full_like(ids0, -100), torch.full_like(ids1, -100)]
        inp1 = um.prepare_multimodal_inputs([ids0], [lab[0]], [mods[0]], ['avqa'])
        inp2 = um.prepare_multimodal_inputs([ids0, ids1], lab, mods, ['avqa', 'avqa'])
        # go through the HF loop exactly as UnifiedForCausalLM.generate does (inputs_embeds only)
        r1 = super(type(um), um).generate(inputs_embeds=inp1["inputs_embeds"], **gen_kw)
        r2 = super(type(um), um).generate(inputs_embeds=inp2["inputs_embeds"], **gen_kw)
        l1, l2 = torch.stack(r1.logits, dim=1), torch.stack(r2.logits, dim=1)
        t1, t2 = l1.topk(2, dim=-1).values, l2.topk(2, dim=-1).values
        margin = min(float((t1[..., 0] - t1[..., 1]).min()), float((t2[..., 0] - t2[..., 1]).min()))
        return margin, (ids0, ids1, mods, lab, inp1, inp2, r1, r2, l1, l2)

    best = None
    for trial in range(24):
        m_, res = run(2 * trial, 2 * trial + 1)
        if best is None or m_ > best[0]:
            best = (m_, res, 2 * trial)
        if m_ > 0.25:
            break
    margin, (ids0, ids1, mods, lab, inp1, inp2, r1, r2, l1, l2), c0 = best
    print(f"chosen clip pair ({c0},{c0 + 1}) min top-2 margin {margin:.4f}; max|logit| {float(l2.abs().max()):.3f}")
    out = {}
    out["embeds_bs1"] = inp1["inputs_embeds"]
    out["embeds_bs2"] = inp2["inputs_embeds"]
    out["pos_bs2"] = inp2["position_ids"]
    out["mask_bs2"] = inp2["attention_mask"]
    out["ids_bs1"] = r1.sequences
    out["logits_bs1"] = l1
    out["ids_bs2"] = r2.sequences
    out["logits_bs2"] = l2
    # the public API too (ids only)
    pub = model.generate(batch_input_ids=[ids0], batch_labels=[lab[0]], batch_X_modals=[mods[0]],
                         batch_task_names=['avqa'], use_cache=True, max_new_tokens=NEW, do_sample=False,
                         pad_token_id=2, eos_token_id=None)
    assert torch.equal(pub, r1.sequences), (pub, r1.sequences)
    print("bs1 ids", r1.sequences.tolist())
    print("bs2 ids", r2.sequences.tolist())

    # plain decoder: prefill logits for all rows + hidden (post-norm) of last layer
    fo = super(type(um), um).forward(inputs_embeds=inp1["inputs_embeds"], output_hidden_states=True, use_cache=False)
    out["prefill_logits_bs1"] = fo.logits
    out["prefill_hidden_bs1"] = fo.hidden_states[-1]

    meta = dict(seed=SEED, dec=TINY_DEC, clip=TINY_CLIP, select=TINY_CLIP_SELECT, beats=TINY_BEATS, qf=TINY_QF,
                d_model=D_MODEL, base_vocab=base_vocab, pad_token_id=2, special=tab, table=table, new_tokens=NEW,
                prompts=dict(n0=24, n1=17, t_v=2, t_a=3, l_a=98, clip0=c0, clip1=c0 + 1), min_margin=margin)
    save("full_tiny_llama", meta, ids0=ids0, ids1=ids1, **out)

    # ---- forward() with the left-pad attention_mask / position_ids HONOURED (models/unified_llama.py:129-160: the multimodal branch of
    # forward() hands prepare_multimodal_inputs' mask and cumsum-1 positions to the decoder, unlike generate()).  Encoders must not run
    # again after a generate() in this process (transformers-5.15 hidden-state recorder artefact, see golden_full_qwen), so the
    # decoder is called exactly as unified_llama.py:149-160 does, on the prepared bs-2 inputs: prefill with the cache kept, then the
    # 1-token decode shortcut (:125-127) with the extended mask and per-row positions.
    mask2, pos2 = inp2["attention_mask"], inp2["position_ids"]
    fm = super(type(um), um).forward(inputs_embeds=inp2["inputs_embeds"], attention_mask=mask2, position_ids=pos2, use_cache=True,
                                     output_hidden_states=True)
    tok = fm.logits[:, -1].argmax(-1)
    mask3 = torch.cat([mask2, torch.ones_like(mask2[:, :1])], 1)
    pos3 = pos2[:, -1:] + 1
    fs = um.forward(input_ids=tok[:, None], attention_mask=mask3, position_ids=pos3, past_key_values=fm.past_key_values, use_cache=True)
    # the same batch WITHOUT the mask (what generate() feeds): shows that the mask matters for the padded row and not for the full one
    fu = super(type(um), um).forward(inputs_embeds=inp2["inputs_embeds"], use_cache=False)
    d_pad = float((fu.logits[1] - fm.logits[1])[mask2[1].bool()].abs().max())
    d_full = float((fu.logits[0] - fm.logits[0]).abs().max())
    print(f"masked forward: padded row differs from the mask-less run by {d_pad:.3e}, full row by {d_full:.3e}")
    assert d_pad > 1e-3 and d_full < 1e-4
    save("forward_masked_tiny_llama", dict(meta, pad_row_maskless_diff=d_pad), ids0=ids0, ids1=ids1, embeds_bs2=inp2["inputs_embeds"],
         mask_bs2=mask2, pos_bs2=pos2, logits_bs2=fm.logits, hidden_bs2=fm.hidden_states[-1], step_tok=tok, step_mask=mask3, step_pos=pos3,
         step_logits=fs.logits)


def golden_id_stats(me, n_clips=24, new=12):
    """UNSEARCHED clips for the id-parity statistic: the tiny full model of `full_tiny_llama` (same seeded weights) on
    clip indices 200 .. 200 + n_clips - 1, one clip per generate() call, no selection of any kind - the reference's greedy ids, per-step
    last-row logits and top-2 margins.  The GPU test reports (does not gate on) the fraction of steps whose id equals the reference's and the
    reference margin at each first divergence."""
    model, cfg = build_full_model(me, TINY_DEC)
    inner = model.get_model()
    inner.pad_token_id = 2
    inner.visual_encoder = build_visual_encoder(me)
    inner.vl_projector = me.VLProjector(hidden_size=128, image_token_nums=256, num_query_token=32, num_hidden_layers=2, d_model=D_MODEL, depth=2)

    class AE(me.AudioEncoder):
        def __init__(self, beats):
            torch.nn.Module.__init__(self)
            self.audio_encoder = beats
    beats = build_beats(TINY_BEATS)
    inner.audio_encoder = AE(beats)
    inner.al_projector = me.ALProjector(hidden_size=128, num_query_token=32, num_hidden_layers=2, d_model=D_MODEL, depth=2)
    tok = _Tok(TINY_DEC["vocab_size"] - 17)
    base_vocab = len(tok)
    model.base_model.model.initialize_MM_tokenizer(tok, mask_token_nums=6, use_vqgan=False)
    model.eval()
    alias = [("base_model.model.model.audio_encoder.audio_encoder." + c, ["base_model.model.model.audio_encoder.audio_encoder." + o for o in os_])
             for c, os_ in beats_alias(beats)]
    table = load_synth(model, "", alias_groups=alias)
    um = model.base_model.model
    tab = dict(um.SPECIAL_TOKEN_2_IDS)
    gen_kw = dict(use_cache=True, max_new_tokens=new, do_sample=False, output_logits=True, return_dict_in_generate=True, pad_token_id=2, eos_token_id=None)
    clips = list(range(200, 200 + n_clips))
    nts = [16 + (c * 7) % 17 for c in clips]                       # prompt lengths 16 .. 32, fixed by the clip index
    preps = []
    for c, nt in zip(clips, nts):                                  # every encoder pass BEFORE the first generate() (transformers-5.15 artefact)
        ids = synth.synth_prompt_ids(nt, base_vocab, tab, seed=SEED, clip=c)
        mods = {'<video>': synth.synth_video(2, seed=SEED, clip=c), '<audio>': synth.synth_audio(3, 98, seed=SEED, clip=c)}
        preps.append(um.prepare_multimodal_inputs([ids], [torch.full_like(ids, -100)], [mods], ['avqa'])["inputs_embeds"])
    ids_out, logits_out = [], []
    for e in preps:
        r = super(type(um), um).generate(inputs_embeds=e, **gen_kw)
        ids_out.append(r.sequences[0])
        logits_out.append(torch.stack(r.logits, dim=1)[0])
    ids_out, logits_out = torch.stack(ids_out), torch.stack(logits_out)
    t2 = logits_out.topk(2, dim=-1).values
    margin = t2[..., 0] - t2[..., 1]
    print("id_stats: margins min / median", float(margin.min()), float(margin.median()), "scale", float(logits_out.abs().max()))
    meta = dict(seed=SEED, dec=TINY_DEC, clip=TINY_CLIP, select=TINY_CLIP_SELECT, beats=TINY_BEATS, qf=TINY_QF, d_model=D_MODEL, base_vocab=base_vocab,
                pad_token_id=2, special=tab, table=table, new_tokens=new, clips=clips, prompt_tokens=nts, t_v=2, t_a=3, l_a=98,
                note="unsearched clip indices: no margin selection")
    save("id_stats_tiny_llama", meta, ids=ids_out, logits=logits_out, margin=margin)


def golden_sharp(me, n_keep=8, new=8, min_margin=0.15, n_cand=800):
    """The ids clause of north_star on fixtures whose margins make it decidable: the tiny full model of `full_tiny_llama` (same
    seeded weights) on the first `n_keep` clips of indices 400.. whose reference top-2 logit margin is >= `min_margin` on EVERY one of the
    `new` greedy steps - 0.15 on a logit scale of ~3.8 is 10 x the logit error a bf16-MFMA implementation of this stack shows (operand floor /
    storage emulation ~4e-3 of the scale = 0.015).  A random decoder's top-2 gap has a median of 0.23 and no lower bound (the unsearched clips of
    id_stats_tiny_llama.npz: the best of 24 has a minimum gap of 0.106 over 12 steps), so such clips are ~2 % of all candidates: the first
    `n_keep` of the clip indices 400, 401, ... that qualify are recorded, and how many were tried.  On these clips greedy ids must be
    EQUAL to the reference's, step for step, with no margin escape.  Stored: clip indices, prompt lengths, ids, per-step last-row logits, margins."""
    model, cfg = build_full_model(me, TINY_DEC)
    inner = model.get_model()
    inner.pad_token_id = 2
    beats = _attach_tiny_encoders(me, inner, D_MODEL)
    tok = _Tok(TINY_DEC["vocab_size"] - 17)
    base_vocab = len(tok)
    model.base_model.model.initialize_MM_tokenizer(tok, mask_token_nums=6, use_vqgan=False)
    model.eval()
    alias = [("base_model.model.model.audio_encoder.audio_encoder." + c, ["base_model.model.model.audio_encoder.audio_encoder." + o for o in os_])
             for c, os_ in beats_alias(beats)]
    table = load_synth(model, "", alias_groups=alias)
    um = model.base_model.model
    tab = dict(um.SPECIAL_TOKEN_2_IDS)
    gen_kw = dict(use_cache=True, max_new_tokens=new, do_sample=False, output_logits=True, return_dict_in_generate=True, pad_token_id=2, eos_token_id=None)
    clips = list(range(400, 400 + n_cand))
    nts = [16 + (c * 5) % 19 for c in clips]
    preps = []
    for c, nt in zip(clips, nts):                                  # every encoder pass BEFORE the first generate() (transformers-5.15 artefact)
        ids = synth.synth_prompt_ids(nt, base_vocab, tab, seed=SEED, clip=c)
        mods = {'<video>': synth.synth_video(2, seed=SEED, clip=c), '<audio>': synth.synth_audio(3, 98, seed=SEED, clip=c)}
        preps.append(um.prepare_multimodal_inputs([ids], [torch.full_like(ids, -100)], [mods], ['avqa'])["inputs_embeds"])
    keep = []
    for i, e in enumerate(preps):
        r = super(type(um), um).generate(inputs_embeds=e, **gen_kw)
        lg = torch.stack(r.logits, dim=1)[0]
        t2 = lg.topk(2, dim=-1).values
        m_ = t2[:, 0] - t2[:, 1]
        if float(m_.min()) >= min_margin:
            keep.append((clips[i], nts[i], r.sequences[0], lg, m_))
            if len(keep) == n_keep:
                break
    assert len(keep) == n_keep, f"only {len(keep)} of {n_cand} candidate clips have every margin >= {min_margin}"
    scale = max(float(k[3].abs().max()) for k in keep)
    print(f"sharp: kept clips {[k[0] for k in keep]} after {i + 1} candidates; min margin {min(float(k[4].min()) for k in keep):.3f}, logit scale {scale:.3f}")
    meta = dict(seed=SEED, dec=TINY_DEC, clip=TINY_CLIP, select=TINY_CLIP_SELECT, beats=TINY_BEATS, qf=TINY_QF, d_model=D_MODEL, base_vocab=base_vocab,
                pad_token_id=2, special=tab, table=table, new_tokens=new, clips=[k[0] for k in keep], prompt_tokens=[k[1] for k in keep], t_v=2, t_a=3, l_a=98,
                min_margin=min_margin, candidates_tried=i + 1, note="clips selected for margins >= 10 x the bf16 logit error on every step: ids must be EQUAL, no escape")
    save("sharp_tiny_llama", meta, ids=torch.stack([k[2] for k in keep]), logits=torch.stack([k[3] for k in keep]), margin=torch.stack([k[4] for k in keep]))


def golden_avs_loop(me, new=8):
    """The pixel-task loop of the reference, looped (a batched fixture made by looping the reference): the tiny full model +
    SegModule, `generate_avs` (models/unified_llama.py:270-361) called ONE SAMPLE AT A TIME like scripts/quick_start.py:270-450 does, on five
    samples - one clip under three tasks (s4 / avss / ms3: one and 71 class planes) whose six re-pointed <mask_i> ids it emits, and two other clips
    (which produce fewer than six mask tokens: the ids-only outcome, with the reference's message).  A random decoder never emits real mask
    tokens, so the six <mask_i> ids are re-pointed at the tokens sample 0 emits at steps 1..6 (as the GPU tests do).  Stored per sample: the ids,
    whether masks came back, strided samples + checksums of the masks.  Every encoder pass runs BEFORE the first generate() (transformers-5.15
    hidden-state recorder artefact, see golden_full_qwen): prepare_multimodal_inputs is evaluated for all samples first and its results are handed
    to generate_avs in the loop - the reference's own function results, in another call order.  The per-step hidden state the reference picks
    (`output.hidden_states[step][-1]`) is checked to be the POST-final-norm state under this transformers version (lm_head of it = the step's logits)."""
    model, cfg = build_full_model(me, TINY_DEC)
    inner = model.get_model()
    inner.pad_token_id = 2
    beats = _attach_tiny_encoders(me, inner, D_MODEL)
    inner.seg_module = me.SegModule(d_model=D_MODEL, vit_image_embedding_dim=128, prompt_embed_dim=256, image_scale_nums=2, mask_decoder_transformer_depth=2,
                                    token_nums_per_scale=3, avs_query_num=300, num_classes=1, query_generator_num_layers=2, image_size=224, patch_size=14,
                                    image_embedding_size=16)
    inner.low_res_mask_size = 112
    tok = _Tok(TINY_DEC["vocab_size"] - 17)
    base_vocab = len(tok)
    model.base_model.model.initialize_MM_tokenizer(tok, mask_token_nums=6, use_vqgan=False)
    model.eval()
    alias = [("base_model.model.model.audio_encoder.audio_encoder." + c, ["base_model.model.model.audio_encoder.audio_encoder." + o for o in os_])
             for c, os_ in beats_alias(beats)]
    table = load_synth(model, "", alias_groups=alias)
    um = model.base_model.model
    sp = um.SPECIAL_TOKEN_2_IDS
    spec = [(3, 24, 's4'), (3, 24, 'avss'), (3, 24, 'ms3'), (5, 21, 's4'), (6, 27, 'avss')]          # (clip, prompt tokens, task)
    samples = []
    for c, nt, task in spec:
        ids = synth.synth_prompt_ids(nt, base_vocab, dict(sp), seed=SEED, clip=c)
        for a_, b_ in (("<video_start>", "<image_start>"), ("<video>", "<image>"), ("<video_end>", "<image_end>")):
            ids[ids == sp[a_]] = sp[b_]
        mods = {'<image>': synth.synth_video(1, seed=SEED, clip=c), '<audio>': synth.synth_audio(3, 98, seed=SEED, clip=c), '<mask>': torch.zeros(1, 224, 224)}
        samples.append(dict(batch_input_ids=[ids], batch_labels=[torch.full_like(ids, -100)], batch_X_modals=[mods], batch_task_names=[task]))
    real_prepare = um.prepare_multimodal_inputs
    prepared = [real_prepare(batch_input_ids=s_["batch_input_ids"], batch_labels=s_["batch_labels"], batch_X_modals=s_["batch_X_modals"],
                             return_multi_scale_features=True, return_gt_mask=True, batch_task_names=s_["batch_task_names"]) for s_ in samples]
    gen_kw = dict(use_cache=True, max_new_tokens=new, do_sample=False, pad_token_id=2, eos_token_id=None)
    r0 = super(type(um), um).generate(inputs_embeds=prepared[0]["inputs_embeds"], output_hidden_states=True, output_logits=True, return_dict_in_generate=True, **gen_kw)
    for st in range(new):                                   # hs[step][-1] is the post-final-norm state: lm_head of its last row reproduces the step's logits
        d_ = float((um.lm_head(r0.hidden_states[st][-1][:, -1]) - r0.logits[st]).abs().max())
        assert d_ < 1e-4, (st, d_)
    row = r0.sequences[0].tolist()
    for i in range(6):
        sp[f'<mask_{i}>'] = row[1 + i]
    out, metas = {}, []
    for i, s_ in enumerate(samples):
        um.prepare_multimodal_inputs = (lambda i=i: (lambda **kw: prepared[i]))()
        try:
            r = um.generate_avs(**s_, **gen_kw)
        finally:
            um.prepare_multimodal_inputs = real_prepare
        out[f"ids_{i}"] = r["output_ids"]
        has = "pred_masks" in r
        m = dict(clip=spec[i][0], prompt_tokens=spec[i][1], task=spec[i][2], has_masks=has)
        if has:
            pm = r["pred_masks"][0]
            m["shape"], m["cks"] = list(pm.shape), synth.checksum(pm)
            out[f"mask_sub_{i}"] = (pm[:, 3::8, 5::8] if pm.shape[0] > 1 else pm[:, 1::2, ::2]).contiguous()
        metas.append(m)
        print("avs_loop sample", i, m["task"], "ids", r["output_ids"][0].tolist(), "masks", has)
    assert sum(m["has_masks"] for m in metas) >= 3 and not all(m["has_masks"] for m in metas), metas
    meta = dict(seed=SEED, dec=TINY_DEC, clip=TINY_CLIP, select=TINY_CLIP_SELECT, beats=TINY_BEATS, qf=TINY_QF, d_model=D_MODEL, base_vocab=base_vocab,
                pad_token_id=2, special=dict(sp), table=table, new_tokens=new, samples=metas, t_a=3, l_a=98,
                note="mask ids re-pointed at the tokens sample 0 emits at steps 1..6")
    save("avs_loop_tiny", meta, **out)


def golden_holes(me):
    """forward() with a 2-D attention_mask that is NOT left padding (HF's mask utilities accept any mask - the padding
    mask is AND-ed with the causal one): the hyper-LoRA tiny Llama decoder alone on seeded embeddings [2, 21, D]; row 0 has two interior
    holes, row 1 three left pads, a hole and a masked LAST key.  Default positions (arange) and, second run, the cumsum-1 positions a
    caller would derive from the mask.  Then the 1-token decode shortcut (models/unified_llama.py:125-127) on the kept cache with the mask
    extended by a visible key.  Rows without any visible key (the left pads of row 1) are undefined and excluded by the tests."""
    model, cfg = build_full_model(me, TINY_DEC)
    model.eval()
    table = load_synth(model, "")
    um = model.base_model.model
    g = torch.Generator().manual_seed(33)
    S = 21
    emb = torch.randn(2, S, TINY_DEC["hidden_size"], generator=g) * 0.5
    mask = torch.ones(2, S, dtype=torch.long)
    mask[0, 4] = 0; mask[0, 11] = 0
    mask[1, :3] = 0; mask[1, 9] = 0; mask[1, S - 1] = 0
    fo = super(type(um), um).forward(inputs_embeds=emb, attention_mask=mask, use_cache=True, output_hidden_states=True)
    pos = (mask.cumsum(-1) - 1).clamp(min=0)
    fp = super(type(um), um).forward(inputs_embeds=emb, attention_mask=mask, position_ids=pos, use_cache=False)
    fu = super(type(um), um).forward(inputs_embeds=emb, use_cache=False)
    seen = mask.cumsum(-1) > 0                                    # query rows with at least one visible key at or before them
    d = float((fu.logits - fo.logits)[seen].abs().max())
    print(f"holes: masked vs mask-less logits differ by {d:.3e} on defined rows; scale {float(fo.logits[seen].abs().max()):.3f}")
    assert d > 1e-2
    tok = fo.logits[:, -1].argmax(-1)
    mask2 = torch.cat([mask, torch.ones_like(mask[:, :1])], 1)
    pos2 = torch.full((2, 1), S, dtype=torch.long)
    fs = um.forward(input_ids=tok[:, None], attention_mask=mask2, position_ids=pos2, past_key_values=fo.past_key_values, use_cache=True)
    save("forward_holes_tiny_llama", dict(seed=SEED, dec=TINY_DEC, table=table, eseed=33, maskless_diff=d),
         embeds=emb, mask=mask, logits=fo.logits, hidden=fo.hidden_states[-1], pos=pos, logits_pos=fp.logits, step_tok=tok, step_mask=mask2,
         step_pos=pos2, step_logits=fs.logits)


def golden_qwen(me):
    model, cfg = build_full_model(me, TINY_QWEN, qwen=True)
    model.eval()
    table = load_synth(model, "")
    g = torch.Generator().manual_seed(21)
    emb = torch.randn(2, 9, TINY_QWEN["hidden_size"], generator=g)
    um = model.base_model.model
    NEW = 10
    r = um.generate(inputs_embeds=emb, use_cache=True, max_new_tokens=NEW, do_sample=False, output_logits=True,
                    return_dict_in_generate=True, pad_token_id=2, eos_token_id=None)
    lg = torch.stack(r.logits, dim=1)
    top2 = lg.topk(2, dim=-1).values
    print("qwen ids", r.sequences.tolist(), "min margin", float((top2[..., 0] - top2[..., 1]).min()))
    save("decoder_tiny_qwen2", dict(seed=SEED, dec=TINY_QWEN, table=table, eseed=21, new_tokens=NEW, qkv_bias=True),
         embeds=emb, ids=r.sequences, logits=lg)


def _attach_tiny_encoders(me, inner, d_model):
    """Tiny CLIP + BEATs + both Q-Former projectors attached by hand (init_multimodal_modules needs checkpoint paths)."""
    inner.visual_encoder = build_visual_encoder(me)
    inner.vl_projector = me.VLProjector(hidden_size=128, image_token_nums=256, num_query_token=32,
                                        num_hidden_layers=2, d_model=d_model, depth=2)

    class AE(me.AudioEncoder):
        def __init__(self, beats):
            torch.nn.Module.__init__(self)
            self.audio_encoder = beats
    beats = build_beats(TINY_BEATS)
    inner.audio_encoder = AE(beats)
    inner.al_projector = me.ALProjector(hidden_size=128, num_query_token=32, num_hidden_layers=2, d_model=d_model, depth=2)
    return beats


def golden_full_qwen(me):
    """BASELINE configs[2] end to end on the class the reference's eval script selects (scripts/finetune/inference_hyper_lora.py:1327,
    `from models.unified_qwen import UnifiedForCausalLM`): encoders -> prepare_multimodal_inputs (models/unified_arch.py:217-406) ->
    hyper-LoRA Qwen2 decoder (GQA 4/2, q/k/v bias, d_model 256 != the encoders' 128, so both projector MLPs change width) -> greedy
    HF loop from inputs_embeds only.  unified_qwen.py's own generate()/forward() cannot run as shipped (stale kwargs),
    so - as SURVEY 8c prescribes - the class is used for construction and its parents' forward / generate, with the Llama file's
    generate semantics.  The same weights are additionally loaded into the IN-TREE models/qwen/modeling_qwen2.py Qwen2ForCausalLM
    (the reference's vendored statement of the arithmetic) and its prefill logits are required to agree with the HF classes."""
    from peft_hyper import LoraConfig, get_peft_model
    from models.unified_qwen import UnifiedForCausalLM
    from transformers import Qwen2Config
    cfg = Qwen2Config(**TINY_QWEN, max_position_embeddings=2048, tie_word_embeddings=False, use_sliding_window=False, attention_dropout=0.0)
    cfg._attn_implementation = "eager"
    base = UnifiedForCausalLM(cfg)
    peft_config = LoraConfig(task_type="CAUSAL_LM", target_modules="q_proj,k_proj,v_proj,o_proj,gate_proj,down_proj,up_proj".split(','),
                             inference_mode=False, r=8, lora_alpha=16, lora_dropout=0.05, lora_nums=3)
    model = get_peft_model(base, peft_config)
    inner = model.get_model()
    inner.pad_token_id = 2
    DQ = TINY_QWEN["hidden_size"]
    beats = _attach_tiny_encoders(me, inner, DQ)
    tok = _Tok(TINY_QWEN["vocab_size"] - 17)
    base_vocab = len(tok)
    model.base_model.model.initialize_MM_tokenizer(tok, mask_token_nums=6, use_vqgan=False)
    model.eval()
    alias = [("base_model.model.model.audio_encoder.audio_encoder." + c, ["base_model.model.model.audio_encoder.audio_encoder." + o for o in os_])
             for c, os_ in beats_alias(beats)]
    table = load_synth(model, "", alias_groups=alias)
    um = model.base_model.model
    tab = dict(um.SPECIAL_TOKEN_2_IDS)
    NEW = 12
    gen_kw = dict(use_cache=True, max_new_tokens=NEW, do_sample=False, output_logits=True, return_dict_in_generate=True, pad_token_id=2,
                  eos_token_id=None)
    from transformers import Qwen2ForCausalLM as HFQwen

    def prepare(c0, c1):
        ids0 = synth.synth_prompt_ids(24, base_vocab, tab, seed=SEED, clip=c0)
        ids1 = synth.synth_prompt_ids(17, base_vocab, tab, seed=SEED, clip=c1)
        mods = [{'<video>': synth.synth_video(2, seed=SEED, clip=c), '<audio>': synth.synth_audio(3, 98, seed=SEED, clip=c)} for c in (c0, c1)]
        lab = [torch.full_like(ids0, -100), torch.full_like(ids1, -100)]
        # transformers 5.15 artefact: after ANY generate() in the process the CLIP tower's `output_hidden_states` recorder returns 13
        # entries instead of 7 for 6 layers (the capture hooks fire twice), so hidden_states[i] selects other tensors than under the
        # pinned 4.37.2.  Every encoder pass therefore runs BEFORE the first generate(), and the tuple length is checked.
        n_hs = len(inner.visual_encoder.vision_tower(mods[0]['<video>'], output_hidden_states=True).hidden_states)
        assert n_hs == TINY_CLIP["num_hidden_layers"] + 1, n_hs
        inp1 = um.prepare_multimodal_inputs([ids0], [lab[0]], [mods[0]], ['avqa'])
        inp2 = um.prepare_multimodal_inputs([ids0, ids1], lab, mods, ['avqa', 'avqa'])
        return ids0, ids1, mods, lab, inp1, inp2

    def run(prep):
        ids0, ids1, mods, lab, inp1, inp2 = prep
        r1 = HFQwen.generate(um, inputs_embeds=inp1["inputs_embeds"], **gen_kw)
        r2 = HFQwen.generate(um, inputs_embeds=inp2["inputs_embeds"], **gen_kw)
        l1, l2 = torch.stack(r1.logits, dim=1), torch.stack(r2.logits, dim=1)
        t1, t2 = l1.topk(2, dim=-1).values, l2.topk(2, dim=-1).values
        margin = min(float((t1[..., 0] - t1[..., 1]).min()), float((t2[..., 0] - t2[..., 1]).min()))
        return margin, (ids0, ids1, mods, lab, inp1, inp2, r1, r2, l1, l2)

    NTRIAL = 80
    preps = [prepare(100 + 2 * t, 101 + 2 * t) for t in range(NTRIAL)]
    best = None
    for trial in range(NTRIAL):
        m_, res = run(preps[trial])
        if best is None or m_ > best[0]:
            best = (m_, res, 100 + 2 * trial)
        if m_ > 0.25:
            break
    margin, (ids0, ids1, mods, lab, inp1, inp2, r1, r2, l1, l2), c0 = best
    print(f"qwen full: clip pair ({c0},{c0 + 1}) min top-2 margin {margin:.4f}; max|logit| {float(l2.abs().max()):.3f}")
    print("bs1 ids", r1.sequences.tolist())
    print("bs2 ids", r2.sequences.tolist())
    fo = HFQwen.forward(um, inputs_embeds=inp1["inputs_embeds"], output_hidden_states=True, use_cache=False)

    # ---- the in-tree vendored statement (models/qwen/modeling_qwen2.py) on the same weights: prefill logits must agree
    intree = _intree_qwen_prefill(um, cfg, inp1["inputs_embeds"])
    d_intree = float((intree - fo.logits).abs().max())
    print(f"in-tree modeling_qwen2 prefill logits vs HF classes: max abs diff {d_intree:.3e}")
    assert d_intree < 2e-4, d_intree

    meta = dict(seed=SEED, dec=TINY_QWEN, clip=TINY_CLIP, select=TINY_CLIP_SELECT, beats=TINY_BEATS, qf=TINY_QF, d_model=DQ,
                base_vocab=base_vocab, pad_token_id=2, special=tab, table=table, new_tokens=NEW, qkv_bias=True,
                prompts=dict(n0=24, n1=17, t_v=2, t_a=3, l_a=98, clip0=c0, clip1=c0 + 1), min_margin=margin,
                intree_vs_hf_prefill_max_abs=d_intree)
    save("full_tiny_qwen", meta, ids0=ids0, ids1=ids1, embeds_bs1=inp1["inputs_embeds"], embeds_bs2=inp2["inputs_embeds"],
         pos_bs2=inp2["position_ids"], mask_bs2=inp2["attention_mask"], ids_bs1=r1.sequences, logits_bs1=l1, ids_bs2=r2.sequences,
         logits_bs2=l2, prefill_logits_bs1=fo.logits, prefill_hidden_bs1=fo.hidden_states[-1], prefill_logits_intree_bs1=intree)


def _intree_layers(cfg_ns, n_layers, hf_layers):
    """Build in-tree Qwen2DecoderLayers (models/qwen/modeling_qwen2.py:712-809, eager Qwen2Attention :202-317) whose seven
    projections are the reference's hyper-LoRA Linears, and copy the weights of the given HF/peft layers into them by name."""
    import models.qwen.modeling_qwen2 as MQ
    from peft_hyper.tuners.lora import Linear as HyperLinear
    layers = []
    for i in range(n_layers):
        layer = MQ.Qwen2DecoderLayer(cfg_ns, i)
        for mod, names in ((layer.self_attn, ("q_proj", "k_proj", "v_proj", "o_proj")), (layer.mlp, ("gate_proj", "up_proj", "down_proj"))):
            for n in names:
                old = getattr(mod, n)
                setattr(mod, n, HyperLinear(old.in_features, old.out_features, r=8, lora_alpha=16, lora_nums=3, lora_dropout=0.05,
                                            bias=old.bias is not None))
        layer.eval()
        if hf_layers is not None:
            sd = {k: v for k, v in hf_layers[i].state_dict().items()}
            r = layer.load_state_dict(sd, strict=False)
            assert not r.unexpected_keys and all("rotary_emb" in k for k in r.missing_keys), r
        layers.append(layer)
    return layers, MQ


class _MiniCache:
    """The two methods the 4.37-era attention calls (update / get_usable_length), per layer."""

    def __init__(self):
        self.k, self.v = {}, {}

    def get_usable_length(self, new_len, layer_idx=0):
        return 0 if layer_idx not in self.k else self.k[layer_idx].shape[-2]

    def update(self, k, v, layer_idx, cache_kwargs=None):
        self.k[layer_idx] = k if layer_idx not in self.k else torch.cat([self.k[layer_idx], k], dim=-2)
        self.v[layer_idx] = v if layer_idx not in self.v else torch.cat([self.v[layer_idx], v], dim=-2)
        return self.k[layer_idx], self.v[layer_idx]


def _qwen_ns(dec):
    import types as _t
    return _t.SimpleNamespace(hidden_size=dec["hidden_size"], intermediate_size=dec["intermediate_size"],
                              num_attention_heads=dec["num_attention_heads"], num_key_value_heads=dec["num_key_value_heads"],
                              max_position_embeddings=2048, rope_theta=dec["rope_theta"], attention_dropout=0.0, hidden_act="silu",
                              rms_norm_eps=dec["rms_norm_eps"], _attn_implementation="eager", use_sliding_window=False,
                              sliding_window=4096, max_window_layers=28)


def _intree_qwen_prefill(um, cfg, embeds):
    """Prefill logits of the in-tree decoder stack on the weights of `um` (the peft-wrapped reference model)."""
    ns = _qwen_ns(TINY_QWEN)
    layers, MQ = _intree_layers(ns, cfg.num_hidden_layers, list(um.model.layers))
    norm = MQ.Qwen2RMSNorm(cfg.hidden_size, eps=cfg.rms_norm_eps)
    norm.weight.data = um.model.norm.weight.data.clone()
    S = embeds.shape[1]
    mask = torch.full((S, S), torch.finfo(torch.float32).min).triu(1)[None, None]
    x = embeds
    for layer in layers:
        x = layer(x, attention_mask=mask, position_ids=torch.arange(S)[None], past_key_value=None, use_cache=False)[0]
    return torch.nn.functional.linear(norm(x), um.lm_head.weight).float()


def golden_qwen_ops():
    """The reference's vendored Qwen2 arithmetic, models/qwen/modeling_qwen2.py: Qwen2RMSNorm (:83-98), Qwen2RotaryEmbedding +
    apply_rotary_pos_emb at theta = 1e6 (:101-172), and one hyper-LoRA Qwen2DecoderLayer with eager GQA attention and q/k/v bias
    (:202-317, :712-809) run as prefill (S = 6) and as a 1-token decode step against the cache it filled."""
    dec = dict(TINY_QWEN)
    ns = _qwen_ns(dec)
    layers, MQ = _intree_layers(ns, 1, None)
    layer = layers[0]
    table = load_synth(layer, "model.layers.0.")
    D, H, Hk = dec["hidden_size"], dec["num_attention_heads"], dec["num_key_value_heads"]
    d = D // H
    g = torch.Generator().manual_seed(SEED + 7)
    norm = MQ.Qwen2RMSNorm(D, eps=dec["rms_norm_eps"])
    norm.weight.data = 1.0 + 0.1 * torch.randn(D, generator=g)
    xn = torch.randn(2, 5, D, generator=g) * 3.0
    yn = norm(xn)
    rot = MQ.Qwen2RotaryEmbedding(d, max_position_embeddings=2048, base=dec["rope_theta"])
    q = torch.randn(1, H, 7, d, generator=g)
    k = torch.randn(1, Hk, 7, d, generator=g)
    pos = torch.tensor([[0, 1, 2, 3, 9, 170, 1400]])
    cos, sin = rot(k, seq_len=1401)
    qr, kr = MQ.apply_rotary_pos_emb(q, k, cos, sin, pos)
    S = 6
    x = torch.randn(1, S, D, generator=g)
    mask = torch.full((S, S), torch.finfo(torch.float32).min).triu(1)[None, None]
    cache = _MiniCache()
    y = layer(x, attention_mask=mask, position_ids=torch.arange(S)[None], past_key_value=cache, use_cache=True)[0]
    x1 = torch.randn(1, 1, D, generator=g)
    y1 = layer(x1, attention_mask=torch.zeros(1, 1, 1, S + 1), position_ids=torch.tensor([[S]]), past_key_value=cache, use_cache=True)[0]
    assert cache.k[0].shape == (1, Hk, S + 1, d)
    save("qwen_ops", dict(seed=SEED, cfg=dec, table=table, qkv_bias=True),
         norm_w=norm.weight.detach(), norm_x=xn, norm_y=yn.detach(), rope_q=q, rope_k=k, rope_pos=pos, rope_q_out=qr, rope_k_out=kr,
         layer_x=x, layer_y=y, layer_x1=x1, layer_y1=y1, cache_k=cache.k[0], cache_v=cache.v[0])


def golden_seg(me):
    D = 128
    seg = me.SegModule(d_model=D, vit_image_embedding_dim=128, prompt_embed_dim=256, image_scale_nums=2,
                       mask_decoder_transformer_depth=2, token_nums_per_scale=3, avs_query_num=300, num_classes=1,
                       query_generator_num_layers=2, image_size=224, patch_size=14, image_embedding_size=16).eval()
    table = load_synth(seg, "model.seg_module.")
    g = torch.Generator().manual_seed(31)
    pred = torch.randn(2, 6, D, generator=g)
    feats = [torch.randn(2, 256, 128, generator=g) for _ in range(2)]
    tasks = ['avss', 's4']
    out = seg(pred_embeddings=pred, multi_scale_image_feature_list=feats, low_res_mask_size=112, gt_mask=None,
              batch_task_names=tasks)['pred_masks']
    print("seg shapes", [tuple(o.shape) for o in out], float(out[0].abs().max()), float(out[1].abs().max()))
    # fixtures stay small: inputs are regenerated from the seed (same generator call order), outputs are strided samples
    # of both masks plus order-sensitive checksums of the full tensors
    save("seg_tiny", dict(seed=SEED, d_model=D, table=table, tasks=tasks, pseed=31,
                          cks=[synth.checksum(out[0]), synth.checksum(out[1])]),
         avss_sub=out[0][:, 3::8, 5::8].contiguous(), s4_sub=out[1][:, 1::2, ::2].contiguous())


FRONTEND_SHAPES = [(224, 224), (320, 480), (180, 150), (500, 333), (37, 91), (224, 398), (1080, 1920)]


def golden_frontend():
    """The reference's image path: `self.video_processor.preprocess(frames, return_tensors='pt')['pixel_values']`
    (dataset/quick_start_dataset.py:315,457) with the CLIPImageProcessor of openai/clip-vit-large-patch14 (the defaults of
    the class: shortest edge 224 BICUBIC, centre crop 224, 1/255, CLIP mean / std).  Stored: for every shape the uint8
    resized + cropped image (processor with do_rescale = do_normalize = False) and, for two shapes, the float output."""
    from PIL import Image
    from transformers import CLIPImageProcessor
    proc = CLIPImageProcessor()
    arrays, meta = {}, dict(shapes=FRONTEND_SHAPES, seed0=700, mean=list(proc.image_mean), std=list(proc.image_std))
    for i, (h, w) in enumerate(FRONTEND_SHAPES):
        img = synth.synth_image(h, w, 700 + i)
        frames = [Image.fromarray(img)]
        u8 = proc.preprocess(frames, do_rescale=False, do_normalize=False, return_tensors='np')['pixel_values'][0]
        assert u8.shape == (3, 224, 224)
        arrays[f"u8_{i}"] = np.asarray(u8).round().astype(np.uint8)
        if i in (0, 1):
            arrays[f"px_{i}"] = proc.preprocess(frames, return_tensors='np')['pixel_values'][0].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "frontend_clip.npz"), meta=json.dumps(meta), **arrays)
    print("frontend_clip.npz", {k: v.shape for k, v in arrays.items()})


TINY_VQ = dict(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=32, z_channels=32, n_embed=64, embed_dim=32)


def golden_vqgan():
    """models/taming_transformer/vqgan.py VQModel (the class MaskEncoder wraps, multimodal_encoder.py:546-601) at a tiny
    configuration: get_codebook_indices / decode_code outputs and the pre-quantisation latents."""
    from models.taming_transformer.vqgan import VQModel
    c = TINY_VQ
    dd = dict(double_z=False, z_channels=c["z_channels"], resolution=c["resolution"], in_channels=3, out_ch=3, ch=c["ch"],
              ch_mult=c["ch_mult"], num_res_blocks=c["num_res_blocks"], attn_resolutions=c["attn_resolutions"], dropout=0.0)
    m = VQModel(ddconfig=dd, lossconfig=None, n_embed=c["n_embed"], embed_dim=c["embed_dim"]).eval()
    table = load_synth(m, "mask_encoder.vqgan.")
    g = torch.Generator().manual_seed(SEED + 77)
    x = torch.randn(2, 3, c["resolution"], c["resolution"], generator=g)
    with torch.no_grad():
        lat = m.quant_conv(m.encoder(x))
        idx = m.get_codebook_indices(x)
        dec = m.decode_code(idx)
    e = m.quantize.embedding.weight
    z = lat.permute(0, 2, 3, 1).reshape(-1, e.shape[1])
    d = (z ** 2).sum(1, keepdim=True) + (e ** 2).sum(1) - 2 * z @ e.t()
    top2 = d.topk(2, dim=1, largest=False).values
    print("vqgan idx", idx.shape, "latent absmax", float(lat.abs().max()), "min margin", float((top2[:, 1] - top2[:, 0]).min()),
          "dec absmax", float(dec.abs().max()))
    save("vqgan_tiny", dict(seed=SEED, cfg={k: (list(v) if isinstance(v, tuple) else v) for k, v in c.items()}, table=table, xseed=SEED + 77),
         latents=lat, indices=idx, margin=(top2[:, 1] - top2[:, 0]).reshape(2, -1), decoded=dec)


HARNESS_TASKS = ["avqa", "ave", "avvp", "arig", "s4", "ms3", "avss", "ref-avs"]


def golden_harness():
    """The eval harness the hot path is called from: UnifiedTestDataset sample construction
    (dataset/quick_start_dataset.py:149-270), __getitem__ for the video + audio tasks (:277-436: chat-template wrap, frame
    sampling, audio windows) and DataCollatorForUnifiedTestDataset (:623-707), driven with an in-memory tokenizer
    (tests/util.py tiny_tokenizer), a stub decord.VideoReader that records the frame indices it is asked for, a stub
    librosa.load returning a ramp, and `preprocess` replaced by a recorder of the waveform windows (the fbank itself is
    pinned separately).  Stored: instruction strings, wrapped prompts, token ids / labels, frame indices, audio windows."""
    import tempfile
    import types as _t
    from tests.util import MM_SPECIAL, tiny_tokenizer
    rec = dict(frames=[], windows=[])

    class _Batch:
        def __init__(self, a):
            self.a = a

        def asnumpy(self):
            return self.a

    class VideoReader:
        def __init__(self, uri, height, width):
            self.n = int(uri.split("_")[-1].split(".")[0])          # "clip_<vlen>.mp4"
            self.h, self.w = height, width

        def __len__(self):
            return self.n

        def get_batch(self, indices):
            rec["frames"].append(list(indices))
            return _Batch(np.stack([np.full((self.h, self.w, 3), i % 256, np.uint8) for i in indices]))

    def _load(path, sr=16000, mono=True, **k):
        n = int(path.split("_")[-1].split(".")[0])                   # "wave_<samples>.wav"
        return (np.arange(n, dtype=np.float32) + 1.0) / n, sr

    for name, attrs in (("decord", dict(VideoReader=VideoReader)), ("librosa", dict(load=_load)), ("cv2", {})):
        m = _t.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    import dataset.quick_start_dataset as QD
    from transformers import CLIPImageProcessor

    def _rec_preprocess(source, *a, **k):
        rec["windows"].append(source[0].numpy().astype(np.float32).copy())
        return torch.zeros(1, 2, 128)

    saved = (QD.preprocess, QD.get_v2_pallete)          # restored below: nothing this generator patches outlives it (`harness metrics` in one process died)
    QD.preprocess = _rec_preprocess
    tok = tiny_tokenizer()
    tok.add_tokens(MM_SPECIAL, special_tokens=True)
    q = "How many instruments are playing?"
    raw = [dict(task="avqa", audio_path="wave_6000.wav", video_path="clip_37.mp4", question=q),
           dict(task="avqa", audio_path="wave_6031.wav", video_path="clip_5.mp4", question=q),
           dict(task="ave", audio_path="wave_1000.wav", video_path="clip_100.mp4"),
           dict(task="avvp", audio_path="wave_1017.wav", video_path="clip_8.mp4"),
           dict(task="arig", audio_path="a.wav", image_path="x/00003.jpg"),
           dict(task="s4", audio_path="a.wav", image_path="x/00002.png", mask_path="m.png"),
           dict(task="ms3", audio_path="a.wav", image_path="x/00001.png", mask_path="m.png"),
           dict(task="avss", audio_path="a.wav", image_path="x/00004.jpg", mask_path="m.png"),
           dict(task="ref-avs", audio_path="a.wav", image_path="x/00000.jpg", mask_path="m.png", exp="The Dog")]
    cwd = os.getcwd()
    out = dict(instructions={}, prompts=[], frames=[], n_windows=[])
    arrays = {}
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "data"))
        json.dump(raw, open(os.path.join(td, "data", "example.json"), "w"))
        os.chdir(td)
        try:
            QD.get_v2_pallete = lambda **k: None                  # avss: palette file lives on the authors' cluster
            for task in HARNESS_TASKS:
                flag = {"ref-avs": "ref_avs_task"}.get(task, task + "_task")
                ds = QD.UnifiedTestDataset(mode="test", tokenizer=tok, video_processor=CLIPImageProcessor(), video_frame_nums=8, **{flag: True})
                out["instructions"][task] = [s["instruction"] for s in ds.samples]
                if task in ("avqa", "ave", "avvp"):
                    inst = [ds[i] for i in range(len(ds))]
                    col = QD.DataCollatorForUnifiedTestDataset(tokenizer=tok)(inst)
                    for i, it in enumerate(inst):
                        k = f"{task}{i}"
                        out["prompts"].append([k, it["instruction"], it["output"]])
                        arrays[k + "_ids"] = col["batch_input_ids"][i].numpy()
                        arrays[k + "_labels"] = col["batch_labels"][i].numpy()
                        arrays[k + "_video_mean"] = col["batch_X_modals"][i]["<video>"].mean(dim=(1, 2, 3)).numpy()
                        assert col["batch_task_names"][i] == task and sorted(col["batch_X_modals"][i]) == ["<audio>", "<video>"]
        finally:
            os.chdir(cwd)
            QD.preprocess, QD.get_v2_pallete = saved
    out["frames"] = rec["frames"]
    out["n_windows"] = [len(w) for w in rec["windows"]]
    for i, w in enumerate(rec["windows"]):
        arrays[f"win_{i}"] = w
    out["decode_ids"] = [[5, 6, 7, 1, 2], [66, 65, 67, 3]]
    out["decoded"] = tok.batch_decode(out["decode_ids"], skip_special_tokens=False)
    save("harness", out, **arrays)
    print("harness:", {k: len(v) for k, v in out["instructions"].items()}, "frames", rec["frames"], "windows", len(rec["windows"]))


def golden_metrics():
    """Segmentation metrics (SURVEY.md 8 f-1 anchors): the reference's utils/avss_utils.py functions on small seeded masks.
    Inputs are stored (small); outputs are the reference's return values, plus the integer counts formed with the reference's own
    tensor expressions (fp32 torch.sigmoid, torch.linspace thresholds) for the threshold sweep."""
    from utils import avss_utils as R
    g = torch.Generator().manual_seed(SEED + 77)
    N, H, W = 4, 40, 56
    pred = torch.randn(N, H, W, generator=g) * 3
    gt = torch.zeros(N, H, W)
    gt[0] = (torch.rand(H, W, generator=g) > 0.7).float()
    gt[2] = ((pred[2] + torch.randn(H, W, generator=g)) > 0.5).float()            # correlated with the prediction
    gt[3] = 1.0                                                                    # image 1 stays black: no object
    out = {"bin_pred": pred, "bin_gt": gt}
    out["iou_all"] = R.mask_iou(pred, gt)
    out["iou_each"] = torch.stack([R.mask_iou(pred[n:n + 1], gt[n:n + 1]) for n in range(N)])
    out["f_all"] = np.float64(R.Eval_Fmeasure(pred, gt))
    out["f_each"] = np.array([R.Eval_Fmeasure(pred[n:n + 1], gt[n:n + 1]) for n in range(N)], np.float64)
    out["f_black_only"] = np.float64(R.Eval_Fmeasure(pred[1:2], gt[1:2]))
    out["s_each"] = torch.stack([R.metric_s_for_null(pred[n:n + 1]) for n in range(N)])
    th = torch.linspace(0, 1 - 1e-10, 255)
    out["thlist"] = th
    sp = torch.sigmoid(pred)
    out["ge_tp"] = torch.stack([torch.stack([((sp[n] >= th[i]).float() * gt[n]).sum() for i in range(255)]) for n in range(N)]).to(torch.int64)
    out["ge_cnt"] = torch.stack([torch.stack([(sp[n] >= th[i]).float().sum() for i in range(255)]) for n in range(N)]).to(torch.int64)
    pr = [R._eval_pr(sp[n], gt[n], 255) for n in range(N)]
    out["prec"] = torch.stack([p for p, _ in pr])
    out["recall"] = torch.stack([r for _, r in pr])

    BF, C, h, w = 3, 7, 24, 40
    cp = torch.randn(BF, C, h, w, generator=g)
    ct = torch.randint(0, C, (BF, h, w), generator=g)
    ct[1][ct[1] == 5] = 2                                                          # class 5 absent from frame 1 ...
    ct[1, :3] = 255                                                                # ... whose first rows carry an out-of-range label
    agree = torch.rand(h, w, generator=g) < 0.8
    ct[2] = torch.where(agree, cp[2].argmax(0), ct[2])                             # frame 2 mostly right
    ct[2, -2:] = -1                                                                # and two rows of a negative (ignore) label
    cp[0, 6] = -10.0                                                               # class 6 never predicted in frame 0
    out["cls_pred"], out["cls_tgt"] = cp, ct
    mi, fs, cc, vid = R.calc_color_miou_fscore(cp, ct, T=1)
    out["cls_miou"], out["cls_fscore"], out["cls_count"], out["cls_vid"] = mi, fs, cc, torch.stack(vid)
    each = [R.calc_color_miou_fscore(cp[f:f + 1], ct[f:f + 1], T=1) for f in range(BF)]
    out["cls_iou_fc"] = torch.stack([e[0] for e in each])
    out["cls_fs_fc"] = torch.stack([e[1] for e in each])
    # the AVSS ground truth side: palette + colour map -> class ids, from the reference's dataset module (decord / librosa / cv2 stubbed: unused here)
    import tempfile
    import types as _t
    for name in ("decord", "librosa", "cv2"):
        if name not in sys.modules:
            m = _t.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            if name == "decord":
                m.VideoReader = object
            sys.modules[name] = m
    import dataset.quick_start_dataset as QD
    from PIL import Image
    with tempfile.TemporaryDirectory() as td:
        json.dump({f"c{i}": i for i in range(71)}, open(os.path.join(td, "label2idx.json"), "w"))
        pal = QD.get_v2_pallete(label_to_idx_path=os.path.join(td, "label2idx.json"), num_cls=71)
    out["v2_pallete"] = np.asarray(pal, np.int64)
    cls = torch.randint(0, 71, (36, 52), generator=g).numpy()
    rgb = np.asarray(pal, np.uint8)[cls]
    off = torch.rand(36, 52, generator=g).numpy() < 0.1
    rgb[off] = torch.randint(0, 256, (int(off.sum()), 3), generator=g).numpy().astype(np.uint8)      # colours outside the table -> label 0
    rgb[0, :4] = (1, 0, 0)                                                                             # one channel off the background colour
    out["color_mask"] = rgb
    out["color_label"] = QD.color_mask_to_label(Image.fromarray(rgb, "RGB"), pal).astype(np.int64)
    save("seg_metrics", {"seed": SEED + 77, "note": "reference utils/avss_utils.py + dataset/quick_start_dataset.py outputs; torch " + torch.__version__}, **out)


def golden_llama_ops():
    """The in-tree statement of the decoder arithmetic, models/modeling_llama.py (HF 4.37.2 as vendored by the reference):
    LlamaRMSNorm (:103-117), LlamaRotaryEmbedding + apply_rotary_pos_emb (:120-236) and one LlamaDecoderLayer with eager
    attention (:289-455, :765-850) run as prefill (S = 6, causal additive mask) and as a 1-token decode step against the
    cache it filled.  The config is a plain namespace (transformers 5.x's LlamaConfig no longer carries rope_theta) and
    the cache a minimal object with the two methods the 4.37 attention calls (update / get_usable_length)."""
    import types as _t
    import transformers.utils.import_utils as iu
    if not hasattr(iu, "is_torch_fx_available"):
        iu.is_torch_fx_available = lambda: False
    import models.modeling_llama as ML
    D, H, I = 128, 2, 256
    cfg = _t.SimpleNamespace(hidden_size=D, intermediate_size=I, num_attention_heads=H, num_key_value_heads=H, max_position_embeddings=64,
                             rope_theta=10000.0, rope_scaling=None, attention_bias=False, attention_dropout=0.0, hidden_act="silu",
                             pretraining_tp=1, rms_norm_eps=1e-5, _attn_implementation="eager")

    class Cache:
        def __init__(self):
            self.k, self.v = None, None

        def get_usable_length(self, new_len, layer_idx=0):
            return 0 if self.k is None else self.k.shape[-2]

        def update(self, k, v, layer_idx, cache_kwargs=None):
            self.k = k if self.k is None else torch.cat([self.k, k], dim=-2)
            self.v = v if self.v is None else torch.cat([self.v, v], dim=-2)
            return self.k, self.v

    g = torch.Generator().manual_seed(SEED + 5)
    # RMSNorm
    norm = ML.LlamaRMSNorm(D, eps=1e-5)
    norm.weight.data = 1.0 + 0.1 * torch.randn(D, generator=g)
    xn = torch.randn(2, 5, D, generator=g) * 3.0
    yn = norm(xn)
    # RoPE
    rot = ML.LlamaRotaryEmbedding(64, max_position_embeddings=64, base=10000.0)
    q = torch.randn(1, H, 7, 64, generator=g)
    k = torch.randn(1, H, 7, 64, generator=g)
    pos = torch.tensor([[0, 1, 2, 3, 9, 17, 40]])
    cos, sin = rot(k, seq_len=41)
    qr, kr = ML.apply_rotary_pos_emb(q, k, cos, sin, pos)
    # one decoder layer: prefill + one decode step
    from peft_hyper.tuners.lora import Linear as HyperLinear
    layer = ML.LlamaDecoderLayer(cfg, 0)
    # the in-tree attention calls its projections with return_route_weight (:389-391,452): they are hyper-LoRA Linears, wrapped
    # like get_peft_model does for all seven (quick_start.py:476-493)
    for mod, names in ((layer.self_attn, ("q_proj", "k_proj", "v_proj", "o_proj")), (layer.mlp, ("gate_proj", "up_proj", "down_proj"))):
        for n in names:
            old = getattr(mod, n)
            setattr(mod, n, HyperLinear(old.in_features, old.out_features, r=8, lora_alpha=16, lora_nums=3, lora_dropout=0.05, bias=False))
    layer.eval()
    table = load_synth(layer, "model.layers.0.")
    S = 6
    x = torch.randn(1, S, D, generator=g)
    mask = torch.full((S, S), torch.finfo(torch.float32).min).triu(1)[None, None]
    cache = Cache()
    with torch.no_grad():
        y = layer(x, attention_mask=mask, position_ids=torch.arange(S)[None], past_key_value=cache, use_cache=True)[0][0]   # (outputs, route weights)
        x1 = torch.randn(1, 1, D, generator=g)
        y1 = layer(x1, attention_mask=torch.zeros(1, 1, 1, S + 1), position_ids=torch.tensor([[S]]), past_key_value=cache, use_cache=True)[0][0]
    assert cache.k.shape == (1, H, S + 1, 64)
    save("llama_ops", dict(seed=SEED, cfg=dict(hidden_size=D, intermediate_size=I, num_attention_heads=H, num_key_value_heads=H,
                                               rms_norm_eps=1e-5, rope_theta=10000.0), table=table),
         norm_w=norm.weight.detach(), norm_x=xn, norm_y=yn.detach(), rope_q=q, rope_k=k, rope_pos=pos, rope_q_out=qr, rope_k_out=kr,
         layer_x=x, layer_y=y, layer_x1=x1, layer_y1=y1, cache_k=cache.k, cache_v=cache.v)



# ------------------------------------------------------------------ full-width, shallow fixtures
# The kernels the benchmark spends its time in are OTHER template instantiations than the tiny fixtures reach (head_dim 128, GQA group 7,
# RoPE at d = 128, the 256 x 256 ring GEMM at K = 11008 / 18944 / 1024, 12 x 64 BEATs heads with the gated bias at n = 48 / 96, 768-wide
# Q-Former layers with 1024-wide cross-attention keys, the SegModule under d_model 4096 / 1024-wide image features).  Each generator below
# runs ONE or a FEW layers of the reference at the real widths on seeded synth weights and stores outputs only (fp32; rows sampled with a
# stride coprime to every tile height so that all tile residues are covered); weights travel as (name, shape, checksum), inputs as seeds.
WIDE_LLAMA = dict(hidden_size=4096, intermediate_size=11008, num_attention_heads=32, num_key_value_heads=32, rms_norm_eps=1e-5,
                  rope_theta=10000.0)
WIDE_QWEN = dict(hidden_size=3584, intermediate_size=18944, num_attention_heads=28, num_key_value_heads=4, rms_norm_eps=1e-6,
                 rope_theta=1000000.0)
# prefill batch x rows per sequence: M >= 2560 rows so that every projection of the layer takes the 256 x 256 ring kernel (csrc/gemm_glds.hip
# ring_chosen), S > 64 for the 128-row flash forward; Llama decodes 20 x 32 heads >= 257 (attn_decode_kernel<128>), Qwen 64 x 4 kv heads >= 256
# (attn_decode_gqa_kernel<128, 7>).  Row samples: a stride coprime to every tile height
WIDE_SHAPE = {"llama": (20, 128, 23), "qwen": (64, 72, 41)}
WIDE_STEPS = 2
WIDE_CLIP = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=4, num_attention_heads=16, image_size=224, patch_size=14,
                 layer_norm_eps=1e-5)
WIDE_CLIP_SELECT = [1, 2, 3]                                          # the last layer is dead, as layer 24 of the real tower is
WIDE_BEATS = dict(TINY_BEATS, embed_dim=512, encoder_embed_dim=768, encoder_ffn_embed_dim=3072, encoder_attention_heads=12, encoder_layers=2)
WIDE_QF = dict(hidden=768, heads=12, inter=3072)                      # bert-base (the checkpoint multimodal_encoder.py:90,192 names)


def bf16_exact(t):
    """Inputs of the full-width fixtures are bf16-REPRESENTABLE fp32 values: the HIP entry points take bf16 activations, so with these the
    reference (fp32 arithmetic on the same values) and the HIP path start from identical numbers - no input-quantisation term in the comparison."""
    return t.to(torch.bfloat16).float()


def wide_inputs(D, seed, B, S, steps=WIDE_STEPS):
    """The decoder-layer inputs of the *_layer_wide fixtures, regenerated by the tests (same generator call order)."""
    g = torch.Generator().manual_seed(seed)
    x = bf16_exact(torch.randn(B, S, D, generator=g))
    xs = [bf16_exact(torch.randn(B, 1, D, generator=g)) for _ in range(steps)]
    return x, xs


def _wide_layer(layer, D, H, Hk, d, name, meta, cache, seed, shape):
    """Prefill of B sequences x S rows through one reference decoder layer, then WIDE_STEPS cached one-token steps."""
    import time
    B, S, stride = shape
    x, xs = wide_inputs(D, seed, B, S)
    mask = torch.full((S, S), torch.finfo(torch.float32).min).triu(1)[None, None].expand(B, 1, S, S)
    pos = torch.arange(S)[None].expand(B, S)
    t0 = time.time()
    out = layer(x, attention_mask=mask, position_ids=pos, past_key_value=cache, use_cache=True)[0]
    y = out[0] if isinstance(out, tuple) else out                      # the in-tree Llama layer returns (hidden, route weights)
    ys = []
    for t, x1 in enumerate(xs):
        o = layer(x1, attention_mask=torch.zeros(B, 1, 1, S + t + 1), position_ids=torch.full((B, 1), S + t), past_key_value=cache, use_cache=True)[0]
        ys.append(o[0] if isinstance(o, tuple) else o)
    print(f"{name}: reference layer ran in {time.time() - t0:.1f} s; |y| max {float(y.abs().max()):.3f}")
    M = B * S
    rows = torch.arange(0, M, stride)
    step_seqs = torch.arange(0, B, 3 if B > 32 else 1)                  # decode-step outputs: every sequence, or every third of a large batch (incl. the last)
    assert int(step_seqs[-1]) == B - 1
    kc, vc = (cache.k[0], cache.v[0]) if isinstance(cache.k, dict) else (cache.k, cache.v)
    assert kc.shape == (B, Hk, S + len(xs), d), kc.shape
    heads = [0, Hk - 1]
    meta = dict(meta, B=B, S=S, steps=len(xs), row_stride=stride, xseed=seed, cache_seq=B - 1, cache_heads=heads)
    save(name, meta, rows=rows, y_rows=y.reshape(M, D)[rows], step_seqs=step_seqs, y_steps=torch.stack([t[step_seqs, 0] for t in ys]),
         cache_k=kc[B - 1, heads], cache_v=vc[B - 1, heads])


def golden_llama_layer_wide():
    """models/modeling_llama.py:765-837 (LlamaDecoderLayer, eager attention :352-465) with its seven projections wrapped as
    peft_hyper/tuners/lora.py:260-369 hyper-LoRA Linears, at Llama-2-7B widths (4096 / 11008, 32 heads x 128)."""
    import types as _t
    import transformers.utils.import_utils as iu
    if not hasattr(iu, "is_torch_fx_available"):
        iu.is_torch_fx_available = lambda: False
    import models.modeling_llama as ML
    from peft_hyper.tuners.lora import Linear as HyperLinear
    c = WIDE_LLAMA
    cfg = _t.SimpleNamespace(**c, max_position_embeddings=256, rope_scaling=None, attention_bias=False, attention_dropout=0.0, hidden_act="silu",
                             pretraining_tp=1, _attn_implementation="eager")
    layer = ML.LlamaDecoderLayer(cfg, 0)
    for mod, names in ((layer.self_attn, ("q_proj", "k_proj", "v_proj", "o_proj")), (layer.mlp, ("gate_proj", "up_proj", "down_proj"))):
        for n in names:
            old = getattr(mod, n)
            setattr(mod, n, HyperLinear(old.in_features, old.out_features, r=8, lora_alpha=16, lora_nums=3, lora_dropout=0.05, bias=False))
    layer.eval()
    table = load_synth(layer, "model.layers.0.")

    class Cache:
        def __init__(self):
            self.k, self.v = None, None

        def get_usable_length(self, new_len, layer_idx=0):
            return 0 if self.k is None else self.k.shape[-2]

        def update(self, k, v, layer_idx, cache_kwargs=None):
            self.k = k if self.k is None else torch.cat([self.k, k], dim=-2)
            self.v = v if self.v is None else torch.cat([self.v, v], dim=-2)
            return self.k, self.v
    _wide_layer(layer, c["hidden_size"], 32, 32, 128, "llama_layer_wide", dict(seed=SEED, cfg=c, table=table), Cache(), SEED + 101, WIDE_SHAPE["llama"])


DECODE_REGIME_SHAPE = (512, 8)                      # sequences x prompt rows of the decode-regime fixture


def decode_regime_seqs(B=DECODE_REGIME_SHAPE[0]):
    """The sequences whose outputs the decode-regime fixture stores: 0 .. 7 (the reference's own eval batch) and every 13th after them - so that a
    test may run ANY first-B' sequences (1, 8, 48, 100, 256, 512: one per kernel regime of the decode projections) and find stored rows among them."""
    return list(range(8)) + list(range(8 + 5, B, 13)) + ([B - 1] if (B - 1 - 13) % 13 else [])


def golden_llama_decode_regimes_wide():
    """ONE reference run that pins every decode regime of the projections (beyond the five instantiations it names): the
    hyper-LoRA Llama-2-7B-wide layer of golden_llama_layer_wide on 512 independent sequences of 8 prompt rows, then ONE cached decode step for all
    512.  Sequences do not interact, so the stored rows of the first B' sequences are the reference values of a B'-row decode step whatever B' is:
    the GPU test decodes B' = 1 / 8 (gemm_skinny_dma_kernel + rowfin tails + the fused small-batch attention), 48 (64-row split-K), 100 (panel
    kernel above the 64-row floor), 256 (gemm_dec_ws_kernel) and 512 rows (gemm_dec2_kernel, the benchmark's regime) against them."""
    import types as _t
    import transformers.utils.import_utils as iu
    if not hasattr(iu, "is_torch_fx_available"):
        iu.is_torch_fx_available = lambda: False
    import models.modeling_llama as ML
    from peft_hyper.tuners.lora import Linear as HyperLinear
    c = WIDE_LLAMA
    cfg = _t.SimpleNamespace(**c, max_position_embeddings=64, rope_scaling=None, attention_bias=False, attention_dropout=0.0, hidden_act="silu",
                             pretraining_tp=1, _attn_implementation="eager")
    layer = ML.LlamaDecoderLayer(cfg, 0)
    for mod, names in ((layer.self_attn, ("q_proj", "k_proj", "v_proj", "o_proj")), (layer.mlp, ("gate_proj", "up_proj", "down_proj"))):
        for n in names:
            old = getattr(mod, n)
            setattr(mod, n, HyperLinear(old.in_features, old.out_features, r=8, lora_alpha=16, lora_nums=3, lora_dropout=0.05, bias=False))
    layer.eval()
    table = load_synth(layer, "model.layers.0.")                     # the same weights as llama_layer_wide.npz (same names, same seed)
    B, S = DECODE_REGIME_SHAPE
    D = c["hidden_size"]
    x, xs = wide_inputs(D, SEED + 103, B, S, steps=1)
    mask = torch.full((S, S), torch.finfo(torch.float32).min).triu(1)[None, None].expand(B, 1, S, S)

    class Cache:
        def __init__(self):
            self.k, self.v = None, None

        def get_usable_length(self, new_len, layer_idx=0):
            return 0 if self.k is None else self.k.shape[-2]

        def update(self, k, v, layer_idx, cache_kwargs=None):
            self.k = k if self.k is None else torch.cat([self.k, k], dim=-2)
            self.v = v if self.v is None else torch.cat([self.v, v], dim=-2)
            return self.k, self.v
    cache = Cache()
    y = layer(x, attention_mask=mask, position_ids=torch.arange(S)[None].expand(B, S), past_key_value=cache, use_cache=True)[0][0]
    y1 = layer(xs[0], attention_mask=torch.zeros(B, 1, 1, S + 1), position_ids=torch.full((B, 1), S), past_key_value=cache, use_cache=True)[0][0]
    seqs = torch.tensor(decode_regime_seqs(B))
    print(f"llama_decode_regimes_wide: {len(seqs)} of {B} sequences stored; |y1| max {float(y1.abs().max()):.3f}")
    save("llama_decode_regimes_wide", dict(seed=SEED, cfg=c, table=table, B=B, S=S, steps=1, xseed=SEED + 103),
         seqs=seqs, y_last=y[seqs, -1], y_step=y1[seqs, 0])


def golden_qwen_layer_wide():
    """models/qwen/modeling_qwen2.py:712-809 (Qwen2DecoderLayer, eager GQA attention with q/k/v bias :202-317) with hyper-LoRA projections, at
    Qwen2-7B widths (3584 / 18944, 28 query heads / 4 kv heads x 128, theta 1e6, eps 1e-6)."""
    c = WIDE_QWEN
    layers, MQ = _intree_layers(_qwen_ns(c), 1, None)
    table = load_synth(layers[0], "model.layers.0.")
    _wide_layer(layers[0], c["hidden_size"], 28, 4, 128, "qwen_layer_wide", dict(seed=SEED, cfg=c, table=table, qkv_bias=True), _MiniCache(), SEED + 102, WIDE_SHAPE["qwen"])


def golden_clip_wide(me):
    """CLIP ViT-L/14 widths (1024 / 16 heads x 64 / 4096, 257 tokens a frame), 4 layers, 36 frames through the reference's VisualEncoder
    (models/multimodal_encoder.py:52-84): patch embedding + pre-LN at width 1024, head_dim 64 attention at S = 257, and M = 9252 rows so that
    all four projections of a layer (K = 1024 and 4096, N = 3072 / 1024 / 4096 / 1024) fill the ring kernel's grid as they do in the benchmark."""
    ve = build_visual_encoder(me, WIDE_CLIP, WIDE_CLIP_SELECT)
    table = load_synth(ve, "model.visual_encoder.")
    T = 36
    video = bf16_exact(synth.synth_video(T, seed=SEED, clip=9)[None])
    feats = ve(video)
    rows = torch.arange(0, feats[0].shape[1], 79)
    save("clip_wide", dict(seed=SEED, cfg=WIDE_CLIP, select=WIDE_CLIP_SELECT, table=table, t_v=T, clip=9, row_stride=79),
         rows=rows, f0=feats[0][0, rows], f1=feats[1][0, rows], f2=feats[2][0, rows])


def golden_beats_wide():
    """BEATs iter3+ widths (patch embed 512 -> 768, 12 heads x 64, FFN 3072, 320 buckets / max distance 800, deep-norm, gated relative position
    bias), 2 encoder layers, at both audio window lengths: L = 98 (n = 48 tokens) and L = 198 (n = 96); models/beats/BEATs.py:134-182,
    models/beats/backbone.py."""
    m = build_beats(WIDE_BEATS)
    table = load_synth(m, "model.audio_encoder.audio_encoder.", alias_groups=beats_alias(m))
    # L = 98: 256 one-second windows = 12288 token rows (the 768-wide GEMMs then run on the ring kernel, as at the benchmark's 256 clips x 10
    # windows), sampled rows stored; L = 198 (the 2-second windows of MUSIC-AVQA, gated bias at n = 96): 3 windows, stored whole
    outs, t_a = {}, {"98": 256, "198": 3}
    for L in (98, 198):
        x = bf16_exact(synth.synth_audio(t_a[str(L)], L, seed=SEED, clip=L + 1))
        y, _ = m.extract_features(x, padding_mask=torch.zeros(x.shape[:-1]).bool(), feature_only=True)
        outs[f"y{L}"] = y
    rows98 = torch.arange(0, outs["y98"].shape[0] * outs["y98"].shape[1], 113)
    outs["rows98"] = rows98
    outs["y98"] = outs["y98"].reshape(-1, outs["y98"].shape[-1])[rows98]
    save("beats_wide", dict(seed=SEED, cfg=WIDE_BEATS, table=table, clips={"98": 99, "198": 199}, t_a=t_a), **outs)


def golden_projectors_wide():
    """Both Q-Former projectors at their real configuration (models/multimodal_encoder.py:87-262 over models/Qformer.py: bert-base layers 768 /
    12 heads / 3072, 2 layers, 32 queries; cross-attention keys 1024 wide (CLIP) and 768 wide (BEATs); output MLP 768 -> 4096 -> 4096)."""
    me = ref_shims.patch_bert_config(lambda: ref_shims.tiny_bert_config(**WIDE_QF))
    vl = me.VLProjector(hidden_size=1024, image_token_nums=256, num_query_token=32, num_hidden_layers=2, d_model=4096, depth=2).eval()
    tv = load_synth(vl, "model.vl_projector.")
    g = torch.Generator().manual_seed(111)
    feat = bf16_exact(torch.randn(1, 2 * 256, 1024, generator=g))
    yv = vl(feat)
    al = me.ALProjector(hidden_size=768, num_query_token=32, num_hidden_layers=2, d_model=4096, depth=2).eval()
    ta = load_synth(al, "model.al_projector.")
    g = torch.Generator().manual_seed(112)
    af = bf16_exact(torch.randn(1, 2, 96, 768, generator=g))
    ya = al(af)
    save("projectors_wide", dict(seed=SEED, qf=WIDE_QF, d_model=4096, table=tv + ta, vseed=111, aseed=112, vshape=[1, 512, 1024], ashape=[1, 2, 96, 768]),
         vout=yv, aout=ya)


def golden_seg_wide(me):
    """SegModule as scripts/quick_start.py:505-529 builds it (models/unified_arch.py:91-106): d_model 4096, CLIP features 1024 wide, prompt
    dim 256, 300 queries, two mask-decoder levels of depth 2; one `avss` (71 classes) and one `s4` sample."""
    D = 4096
    seg = me.SegModule(d_model=D, vit_image_embedding_dim=1024, prompt_embed_dim=256, image_scale_nums=2, mask_decoder_transformer_depth=2,
                       token_nums_per_scale=3, avs_query_num=300, num_classes=1, query_generator_num_layers=2, image_size=224, patch_size=14,
                       image_embedding_size=16).eval()
    table = load_synth(seg, "model.seg_module.")
    g = torch.Generator().manual_seed(131)
    pred = bf16_exact(torch.randn(2, 6, D, generator=g))
    feats = [bf16_exact(torch.randn(2, 256, 1024, generator=g)) for _ in range(2)]
    tasks = ['avss', 's4']
    out = seg(pred_embeddings=pred, multi_scale_image_feature_list=feats, low_res_mask_size=112, gt_mask=None, batch_task_names=tasks)['pred_masks']
    print("seg_wide shapes", [tuple(o.shape) for o in out], float(out[0].abs().max()), float(out[1].abs().max()))
    save("seg_wide", dict(seed=SEED, d_model=D, vit_dim=1024, table=table, tasks=tasks, pseed=131, cks=[synth.checksum(out[0]), synth.checksum(out[1])]),
         avss_sub=out[0][:, 3::8, 5::8].contiguous(), s4_sub=out[1][:, 1::2, ::2].contiguous())


def main():
    """No argument: every generator of tests/golden/registry.py, each in its own interpreter (order-independent by construction).  One name: that
    generator in this process.  Several names / a group name (`fullwidth`): one interpreter each."""
    import inspect
    import subprocess
    from registry import GENERATORS, GROUPS
    names = []
    for a in sys.argv[1:] or list(GENERATORS):
        names += GROUPS.get(a, [a])
    unknown = [n for n in names if n not in GENERATORS]
    if unknown:
        raise SystemExit(f"unknown generator(s) {unknown}; known: {sorted(GENERATORS)} + groups {sorted(GROUPS)}")
    if len(names) > 1:
        failed = []
        for n in names:
            print(f"==== {n}", flush=True)
            kind, entry, _ = GENERATORS[n]
            cmd = [sys.executable, os.path.join(HERE, entry)] if kind.startswith("script") else [sys.executable, os.path.abspath(__file__), n]
            rc = subprocess.run(cmd).returncode
            if rc != 0 and not (kind == "script_optional" and rc == 3):
                failed.append(n)
        if failed:
            raise SystemExit(f"generators failed: {failed}")
        return
    kind, entry, _ = GENERATORS[names[0]]
    if kind.startswith("script"):
        raise SystemExit(subprocess.run([sys.executable, os.path.join(HERE, entry)]).returncode)
    ref_shims.install()
    me = ref_shims.patch_bert_config(lambda: ref_shims.tiny_bert_config(**TINY_QF))
    fn = globals()[entry]
    fn(me) if inspect.signature(fn).parameters else fn()


if __name__ == "__main__":
    main()
