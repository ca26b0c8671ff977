"""Differential fuzz of prepare_multimodal_inputs (CLIP tower -> VLProjector / BEATs -> ALProjector -> splice -> left pad) on RANDOM tiny encoder
configurations against the fp32 CPU oracle: widths, depths, head counts, selected CLIP levels, frames, audio segments and window lengths, ragged
prompts.  The bound is max(REL, 2.5 x the oracle's own bf16-storage emulation on the same configuration).
    python scripts/fuzz_multimodal.py [configs] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import synth
from oracle import crab_oracle as O
from tests.util import build_tiny_crab

BF = torch.bfloat16
NCFG = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
REL = 1.4e-2
bad, worst = [], 0.0
for ci in range(NCFG):
    dm = rng.choice([128, 256])
    ch, bh, qh = rng.choice([2, 3, 4]), rng.choice([2, 4]), rng.choice([2, 3])      # BEATs width / 16 groups must be a multiple of 8 (stated limit)
    cl = rng.choice([3, 4, 6])
    sel = sorted(rng.sample(range(1, cl + 1), 3)) if cl >= 3 else [1, 2, 3]
    qwen = rng.random() < 0.3
    meta = dict(
        dec=dict(hidden_size=dm, intermediate_size=rng.choice([136, 256]), num_hidden_layers=1, num_attention_heads=dm // 64, num_key_value_heads=dm // 64,
                 vocab_size=320, rms_norm_eps=1e-5, rope_theta=10000.0),
        clip=dict(hidden_size=64 * ch, intermediate_size=rng.choice([136, 256, 520]), num_hidden_layers=cl, num_attention_heads=ch, image_size=224, patch_size=14,
                  layer_norm_eps=1e-5),
        select=sel,
        beats=dict(input_patch_size=16, embed_dim=rng.choice([64, 128]), encoder_embed_dim=64 * bh, encoder_ffn_embed_dim=rng.choice([136, 256]),
                   encoder_attention_heads=bh, encoder_layers=rng.choice([1, 2, 3]), conv_pos=128, conv_pos_groups=16, num_buckets=320, max_distance=800,
                   deep_norm=True, gru_rel_pos=True, conv_bias=False, relative_position_embedding=True, layer_norm_first=False, activation_fn="gelu",
                   dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0, finetuned_model=False),
        qf=dict(hidden=64 * qh, heads=qh, inter=rng.choice([136, 256])), d_model=dm, base_vocab=303, pad_token_id=2, qkv_bias=qwen)
    desc = f"cfg {ci}: d_model={dm} clip={64 * ch}x{cl} sel={sel} beats={64 * bh}x{meta['beats']['encoder_layers']} qf={64 * qh} {'qwen' if qwen else 'llama'}"
    try:
        torch.manual_seed(500 + ci)
        model = build_tiny_crab(meta)
        sd = model.state_dict()
        W = {}
        for k, v in sd.items():
            if not v.dtype.is_floating_point:
                W[k] = v.clone().cpu(); continue
            if v.dim() > 1:
                fan = v[0].numel()
                t = torch.randn(v.shape) * min(0.2, 1.2 / fan ** 0.5)
            elif k.endswith("weight") and ("norm" in k.lower() or "ln" in k.lower().split(".")[-2] or "layer_norm" in k.lower()):
                t = 1 + 0.1 * torch.randn(v.shape)
            else:
                t = 0.05 * torch.randn(v.shape)
            if k.endswith("weight_g"): t = t.abs() + 0.5
            W[k] = t.to(BF).float()
        for k in list(W):                                      # BEATs: every layer aliases layer 0's relative-position table (backbone.py:78-81)
            if k.endswith("self_attn.relative_attention_bias.weight") and ".layers.0." not in k:
                W[k] = W[k.split(".layers.")[0] + ".layers.0.self_attn.relative_attention_bias.weight"]
        r = model.load_state_dict(W, strict=False)
        assert not r.missing_keys, r.missing_keys[:4]
        Wo = O.strip_peft_prefix(W)
        keys = O.BeatsConfig.__dataclass_fields__.keys()
        ocfg = O.CrabConfig(decoder=O.DecoderConfig(**meta["dec"]), clip=O.ClipConfig(**meta["clip"], select_layers=tuple(sel)),
                            beats=O.BeatsConfig(**{k: v for k, v in meta["beats"].items() if k in keys}),
                            qformer=O.QFormerConfig(hidden_size=64 * qh, num_attention_heads=qh, intermediate_size=meta["qf"]["inter"]),
                            base_vocab=303, pad_token_id=2)
        um = model.base_model.model
        B = rng.choice([1, 2, 3])
        tv, ta, la = rng.choice([1, 2, 3, 5]), rng.choice([1, 2, 3, 7]), rng.choice([98, 198, 98, 198, 16, 47, 130, 400])      # r06: window lengths off the two the datasets use
        ids = [synth.synth_prompt_ids(12 + 5 * i + rng.randrange(4), 303, um.SPECIAL_TOKEN_2_IDS, seed=ci, clip=i) for i in range(B)]
        mods = [{'<video>': synth.synth_video(tv, seed=ci, clip=i), '<audio>': synth.synth_audio(ta, la, seed=ci, clip=i)} for i in range(B)]
        lab = [torch.full_like(i, -100) for i in ids]
        got = um.prepare_multimodal_inputs(ids, lab, mods, ['avqa'] * B)
        ref = O.prepare_multimodal_inputs(ids, mods, Wo, ocfg, None)
        emu = O.prepare_multimodal_inputs(ids, mods, Wo, ocfg, BF)
        g, rf, em = got["inputs_embeds"].float().cpu(), ref["inputs_embeds"], emu["inputs_embeds"]
        assert g.shape == rf.shape, (g.shape, rf.shape)
        assert torch.equal(got["attention_mask"].cpu().long(), ref["attention_mask"].long()) and torch.equal(got["position_ids"].cpu().long(), ref["position_ids"].long())
        scale = float(rf.abs().max())
        e, ee = float((g - rf).abs().max()) / scale, float((em - rf).abs().max()) / scale
        worst = max(worst, e)
        print(f"{desc} B={B} t_v={tv} t_a={ta} L_a={la}: rel err {e:.3e} (bf16-storage emulation {ee:.3e})", flush=True)
        if not (e < max(REL, 2.5 * ee)) or not torch.isfinite(g).all():
            bad.append(desc + f" -> inputs_embeds rel err {e:.3e}, emulation {ee:.3e}")
        del model
        torch.cuda.empty_cache()
    except Exception as ex:      # noqa: BLE001
        import traceback
        bad.append(desc + f" -> {type(ex).__name__}: {str(ex)[:300]}")
        traceback.print_exc()
print(f"worst rel err {worst:.3e}; {len(bad)} failures")
for b_ in bad[:30]: print("FAIL", b_)
sys.exit(1 if bad else 0)
