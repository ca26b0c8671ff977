"""Full-size checkpoint-shape coverage: crab_amd's modules against the reference's own state-dict keys and shapes
(tests/golden/ckpt_manifest.npz, written by tests/golden/make_ckpt_manifest.py from the reference classes built at FULL size on the meta
device the way scripts/quick_start.py:505-554 builds them).  On the meta device here (no memory, no GPU); tests/test_fullsize_gpu.py does the
real load_state_dict(strict=True) of synthesised tensors on the GPU box."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# reference keys that crab_amd consumes WITHOUT keeping (dead on the inference path; crab_amd/multimodal_encoder.py documents each)
DEAD = ("Qformer.cls.", "Qformer.bert.embeddings.position_ids", "audio_encoder.audio_encoder.predictor.")


def manifest():
    return json.loads(bytes(np.load(os.path.join(GOLDEN, "ckpt_manifest.npz"))["manifest"]).decode())


def _is_dead(k: str) -> bool:
    return any(d in k for d in DEAD) or (".self_attn.relative_attention_bias.weight" in k and ".encoder.layers.0." not in k)


def canon(k: str) -> str:
    """transformers 5.x flattens CLIPVisionModel's `vision_model.` level away; 4.37.2 (the reference's pin) and the hub files keep it."""
    t = ".visual_encoder.vision_tower."
    if t in k and not k.split(t, 1)[1].startswith("vision_model."):
        a, b = k.split(t, 1)
        return a + t + "vision_model." + b
    return k


@pytest.mark.parametrize("llm", ["llama", "qwen"])
def test_full_size_state_dict_keys_and_shapes_match_the_reference(llm):
    from crab_amd.build_model import build_crab
    ref = manifest()[llm]["state_dict"]
    model = build_crab(llm, device=torch.device("meta"), randomize=False, segment=True)
    ours = {k: list(v.shape) for k, v in model.state_dict().items()}
    live = {canon(k): v for k, v in ref.items() if not _is_dead(k)}
    missing = sorted(set(live) - set(ours))
    extra = sorted(set(ours) - set(live))
    assert not missing, f"{len(missing)} reference keys have no parameter here, e.g. {missing[:5]}"
    assert not extra, f"{len(extra)} parameters here have no reference key, e.g. {extra[:5]}"
    bad = [(k, live[k], ours[k]) for k in live if live[k] != ours[k]]
    assert not bad, bad[:5]
    assert ours["base_model.model.model.embed_tokens.weight"][0] == ours["base_model.model.lm_head.weight"][0] == {"llama": 32017, "qwen": 152081}[llm]
    # the dead keys are exactly the documented families (nothing else is silently dropped)
    dead = [k for k in ref if _is_dead(k)]
    assert len(dead) == 2 * 8 + 11 + 2, len(dead)          # 2 x (7 cls.* + position_ids), 11 aliased bias tables, predictor weight + bias


@pytest.mark.parametrize("llm", ["llama", "qwen"])
def test_finetune_weights_files_are_fully_consumed(llm):
    """`finetune_weights.bin` of the hyper-LoRA stage (vl_projector, al_projector, lora) and of the AVS stage (seg_module, embed_tokens,
    lm_head): every key the trainer writes (scripts/pretrain/trainer.py:183-197) is either a parameter here or a documented dead key."""
    from crab_amd.build_model import build_crab
    m = manifest()[llm]
    model = build_crab(llm, device=torch.device("meta"), randomize=False, segment=True)
    ours = {k: list(v.shape) for k, v in model.state_dict().items()}
    for name in ("finetune_hyperlora", "finetune_avs"):
        f = m[name]
        assert len(f) > 200
        for k, shp in f.items():
            if _is_dead(k):
                continue
            assert canon(k) in ours and ours[canon(k)] == shp, (name, k, shp, ours.get(canon(k)))
    assert sum(".lora_" in k for k in m["finetune_hyperlora"]) == {"llama": 32, "qwen": 28}[llm] * 7 * 5       # route, A, B0..B2 per projection


def test_load_state_dict_strict_consumes_the_reference_key_forms_tiny():
    """strict=True on the CPU with real (tiny) tensors: the 5.x-flattened CLIP keys, `position_ids`, the Q-Former LM head, the BEATs predictor
    and the aliased bias tables all load without a missing or unexpected key."""
    from tests.util import bert_cfg, build_tiny_crab, load_fixture
    meta, _ = load_fixture("full_tiny_llama")
    model = build_tiny_crab(meta, device="cpu")
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    t = "base_model.model.model.visual_encoder.vision_tower."
    flat = {}
    for k, v in sd.items():
        flat[k.replace(t + "vision_model.", t) if k.startswith(t + "vision_model.") else k] = v
    flat[t + "embeddings.position_ids"] = torch.arange(257)[None]
    for q, pre in (("visual_Qformer", "vl_projector"), ("audio_Qformer", "al_projector")):
        p = f"base_model.model.model.{pre}.{q}."
        flat[p + "cls.predictions.bias"] = torch.zeros(7)
        flat[p + "cls.predictions.decoder.weight"] = torch.zeros(7, 3)
        flat[p + "bert.embeddings.position_ids"] = torch.arange(5)[None]
    b = "base_model.model.model.audio_encoder.audio_encoder."
    flat[b + "predictor.weight"], flat[b + "predictor.bias"] = torch.zeros(527, 4), torch.zeros(527)
    flat[b + "encoder.layers.1.self_attn.relative_attention_bias.weight"] = sd[b + "encoder.layers.0.self_attn.relative_attention_bias.weight"].clone()
    r = model.load_state_dict(flat, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    r = model.load_state_dict(sd, strict=True)                     # and the 4.37-style form it writes itself
    assert not r.missing_keys and not r.unexpected_keys


def test_beats_checkpoint_file_cfg_drives_the_module_and_every_key_loads(tmp_path):
    """models/multimodal_encoder.py:157-161: AudioEncoder(ckpt_path) reads {'cfg', 'model'} from the BEATs checkpoint file; `cfg` decides every
    shape (nothing about iter3+ is hard-wired).  A file synthesised with the reference's own keys / shapes at the published iter3+ cfg
    (fine-tuned: with the 527-class predictor) builds the module and loads with strict=True; a second cfg (other widths, no deep-norm, conv
    bias) builds different shapes from the same code."""
    from crab_amd.multimodal_encoder import AudioEncoder
    m = manifest()
    cfg, shapes = m["beats_cfg"], m["llama"]["beats_ckpt_model"]
    g = torch.Generator().manual_seed(0)
    sd = {k: torch.randn(*s, generator=g) * 0.02 for k, s in shapes.items()}
    for i in range(1, cfg["encoder_layers"]):            # the checkpoint stores the aliased table once per layer (backbone.py:78-81)
        sd[f"encoder.layers.{i}.self_attn.relative_attention_bias.weight"] = sd["encoder.layers.0.self_attn.relative_attention_bias.weight"]
    path = os.path.join(tmp_path, "BEATs_iter3_plus_AS2M_finetuned_on_AS2M_cpt2.pt")
    torch.save({"cfg": cfg, "model": sd}, path)
    ae = AudioEncoder(ckpt_path=path, device="cpu")
    ours = {k: list(v.shape) for k, v in ae.audio_encoder.state_dict().items()}
    live = {k: s for k, s in shapes.items() if not k.startswith("predictor.") and not (k.endswith("relative_attention_bias.weight") and ".layers.0." not in k)}
    assert ours == live
    r = ae.audio_encoder.load_state_dict({k: v.to(torch.bfloat16) for k, v in sd.items()}, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    got = ae.audio_encoder.state_dict()["post_extract_proj.weight"]
    assert torch.equal(got, sd["post_extract_proj.weight"].to(torch.bfloat16))                # the constructor really loaded the file
    c = ae.audio_encoder.cfg
    assert (c.encoder_layers, c.encoder_embed_dim, c.num_buckets, c.max_distance, c.deep_norm, c.conv_pos, c.conv_pos_groups) == (12, 768, 320, 800, True, 128, 16)
    other = dict(cfg, encoder_layers=3, encoder_embed_dim=256, encoder_ffn_embed_dim=512, encoder_attention_heads=4, embed_dim=64, deep_norm=False,
                 conv_bias=True, conv_pos=32, conv_pos_groups=4, num_buckets=64, max_distance=128, finetuned_model=False)
    torch.save({"cfg": other, "model": {}}, path)
    small = AudioEncoder(ckpt_path=path, device="cpu").audio_encoder.state_dict()
    assert small["encoder.layers.2.fc1.weight"].shape == (512, 256) and small["patch_embedding.bias"].shape == (64,)
    assert small["encoder.layers.0.self_attn.relative_attention_bias.weight"].shape == (64, 4) and small["encoder.pos_conv.0.weight_v"].shape == (256, 64, 32)


@pytest.mark.gpu
@pytest.mark.parametrize("llm", ["llama", "qwen"])
def test_full_size_reference_checkpoint_loads_strict_on_the_device(llm):
    """The real thing at FULL size on the GPU box: a state dict synthesised with the reference's exact key names and shapes (Llama-2-7B with
    32 017 embedding rows / Qwen2-7B with 152 081, CLIP ViT-L/14, BEATs iter3+, both Q-Formers, SegModule; the manifest's dead keys included)
    loads with strict=True - no missing, no unexpected key - and lands in the packed operands the kernels read; then the two
    finetune_weights.bin key sets load with strict=False (scripts/quick_start.py:537-554) without an unexpected key."""
    from crab_amd.build_model import build_crab
    m = manifest()[llm]
    model = build_crab(llm, randomize=False, segment=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    sd = {}
    for k, shp in m["state_dict"].items():
        if k.endswith("position_ids"):
            sd[k] = torch.arange(shp[-1], device="cuda")[None]
        else:
            sd[k] = (torch.randn(*shp, device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    r = model.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    um = model.base_model.model
    L = um.model.layers[3]
    p = "base_model.model.model.layers.3."
    d = um.config.hidden_size // um.config.num_attention_heads
    Hq = um.config.num_attention_heads * d
    assert torch.equal(L.self_attn._qkv.W[:Hq], sd[p + "self_attn.q_proj.weight"]) and torch.equal(L.self_attn._qkv.W[Hq:Hq + um.config.num_key_value_heads * d], sd[p + "self_attn.k_proj.weight"])
    assert torch.equal(L.mlp._gu.W[0::2], sd[p + "mlp.gate_proj.weight"]) and torch.equal(L.mlp._gu.W[1::2], sd[p + "mlp.up_proj.weight"])       # interleaved rows
    assert torch.equal(L.mlp._down.RA[:3], sd[p + "mlp.down_proj.lora_route.weight"]) and torch.equal(L.mlp._down.RA[3:11], sd[p + "mlp.down_proj.lora_A.weight"])
    assert torch.equal(L.mlp._down.B2[:, 8:16], sd[p + "mlp.down_proj.lora_B1.weight"])
    assert torch.equal(um.lm_head.weight, sd["base_model.model.lm_head.weight"]) and um.lm_head.weight.shape[0] == {"llama": 32017, "qwen": 152081}[llm]
    for name in ("finetune_hyperlora", "finetune_avs"):
        f = {k: sd[k] + 1 for k in m[name]}
        r = model.load_state_dict(f, strict=False)
        assert not r.unexpected_keys, (name, r.unexpected_keys[:4])
        live = [k for k in f if not _is_dead(k)]
        assert set(r.missing_keys).isdisjoint(canon(k) for k in live)
        k0 = live[len(live) // 2]
        assert torch.equal(model.state_dict()[canon(k0)], f[k0])
