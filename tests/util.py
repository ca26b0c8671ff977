"""Shared helpers for the parity tests: fixture loading and weight regeneration."""
from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import numpy as np
import torch

from crab_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# every GPU parity comparison appends {test, what, max_abs, scale, rel, tol} here (gpurun merges gpurun_out/ back; the copy the
# tolerances were set from is committed as profiles/r02_parity_report.json).  The file is reset at session start (conftest.py).
PARITY_REPORT = os.environ.get("CRAB_PARITY_REPORT", os.path.join(ROOT, "gpurun_out", "parity_report.json"))


def record_parity(what: str, max_abs: float, scale: float, tol=None, **extra):
    """Append one measured error to the parity report; never raises (a read-only tree must not fail a parity test)."""
    try:
        test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
        row = {"test": test, "what": what, "max_abs": float(max_abs), "scale": float(scale),
               "rel": float(max_abs) / (float(scale) + 1e-30), "tol_rel": tol}
        row.update(extra)
        os.makedirs(os.path.dirname(PARITY_REPORT), exist_ok=True)
        rows = []
        if os.path.exists(PARITY_REPORT):
            with open(PARITY_REPORT) as f:
                rows = json.load(f)
        rows.append(row)
        with open(PARITY_REPORT, "w") as f:
            json.dump(rows, f, indent=0)
    except Exception:
        pass


def rel_err(got: torch.Tensor, ref: torch.Tensor, what: str = "", tol=None) -> float:
    """max |got - ref| / max |ref| (shapes must match, got must be finite); recorded in the parity report."""
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite output"
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item()
    record_parity(what, err, scale, tol)
    return err / (scale + 1e-9)


_RMS_KEYS = ("input_layernorm.weight", "post_attention_layernorm.weight", "model.norm.weight")


def stored_params(W: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """A checkpoint as the HIP modules HOLD it, in fp32 for the oracle's emulation runs: every floating tensor rounded to bf16 except the
    LayerNorm weights / biases of the encoders and the decoder's RMSNorm weights, which crab_amd keeps in fp32 (crab_amd.ops.NORM_PARAMS_FP32 /
    RMS_DTYPE; the VQGAN's GroupNorms stay bf16).  A LayerNorm weight is recognised structurally - a 1-D `.weight` - and its bias by the sibling name."""
    from crab_amd import ops
    rms_fp32 = ops.RMS_DTYPE == torch.float32                  # r05: the decoder's RMSNorm weights are fp32 beside the fp32 residual stream
    ln_w = {k for k, v in W.items() if k.endswith(".weight") and v.dim() == 1 and (rms_fp32 or not k.endswith(_RMS_KEYS)) and "mask_encoder" not in k}
    out = {}
    for k, v in W.items():
        if not v.is_floating_point():
            out[k] = v
            continue
        keep = ops.NORM_PARAMS_FP32 and (k in ln_w or (k.endswith(".bias") and k[:-5] + ".weight" in ln_w))
        out[k] = v.float() if keep else v.to(torch.bfloat16).float()
    return out


def load_fixture(name: str) -> Tuple[dict, Dict[str, torch.Tensor]]:
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    arrs = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrs


def weights_from_table(meta: dict, verify: bool = True) -> Dict[str, torch.Tensor]:
    """Regenerate the synthetic checkpoint a fixture was made with; checks every checksum."""
    W = {}
    seed = meta["seed"]
    for name, shape, cks in meta["table"]:
        W[name] = synth.synth_tensor(name, shape, seed)
    # tied BEATs table: every layer aliases layer 0's relative_attention_bias (backbone.py:78-81)
    for name in list(W):
        if name.endswith("self_attn.relative_attention_bias.weight") and ".layers.0." not in name:
            pre = name.split(".layers.")[0]
            W[name] = W[pre + ".layers.0.self_attn.relative_attention_bias.weight"]
    if verify:
        for name, shape, cks in meta["table"]:
            got = synth.checksum(W[name])
            assert abs(got - cks) <= 1e-6 * max(1.0, abs(cks)), f"weight regeneration drifted for {name}"
    return W


def strip(W: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in W.items()}


class DuckTokenizer:
    """len()/add_tokens() is all initialize_MM_tokenizer needs (unified_arch.py:409-459)."""

    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def add_tokens(self, toks, special_tokens=False):
        self.n += len(toks)
        return len(toks)


def bert_cfg(qf: dict) -> dict:
    return dict(hidden_size=qf["hidden"], num_attention_heads=qf["heads"], intermediate_size=qf["inter"],
                layer_norm_eps=1e-12, vocab_size=64, max_position_embeddings=64)


def build_tiny_crab(meta: dict, device="cuda"):
    """The product model at the fixture's tiny configuration, wrapped like scripts/quick_start.py:465-529 does
    (models.unified_qwen classes when the fixture says qkv_bias, as inference_hyper_lora.py:1326-1335 selects them)."""
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    if meta.get("qkv_bias"):
        from crab_amd.unified_qwen import UnifiedConfig, UnifiedForCausalLM
        cfg = UnifiedConfig(**meta["dec"], attention_bias=True, pad_token_id=meta["pad_token_id"])
    else:
        from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
        cfg = UnifiedConfig(**meta["dec"], pad_token_id=meta["pad_token_id"])
    cfg.vocab_size = meta["base_vocab"]
    model = get_peft_model(UnifiedForCausalLM(cfg, device=device), LoraConfig())
    model.get_model().pad_token_id = meta["pad_token_id"]
    model.get_model().init_multimodal_modules(d_model=meta["d_model"], visual_branch=True, audio_branch=True,
                                              select_layer_list=meta["select"], clip_config=meta["clip"],
                                              beats_config=meta["beats"], bert_config=bert_cfg(meta["qf"]))
    model.initialize_MM_tokenizer(DuckTokenizer(meta["base_vocab"]), mask_token_nums=6)
    return model


def seg_inputs(meta):
    """Inputs of the seg_tiny fixture, regenerated with the generator call order of make_golden.golden_seg."""
    g = torch.Generator().manual_seed(meta["pseed"])
    D = meta["d_model"]
    pred = torch.randn(2, 6, D, generator=g)
    feats = [torch.randn(2, 256, 128, generator=g) for _ in range(2)]
    return pred, feats


def wide_layer_inputs(meta):
    """Inputs of the llama_layer_wide / qwen_layer_wide fixtures (tests/golden/make_golden.py wide_inputs: same generator call order):
    prefill rows [B, S, D] and the `steps` one-token decode inputs [B, 1, D]."""
    g = torch.Generator().manual_seed(meta["xseed"])
    D = meta["cfg"]["hidden_size"]
    x = bf16_exact(torch.randn(meta["B"], meta["S"], D, generator=g))
    xs = [bf16_exact(torch.randn(meta["B"], 1, D, generator=g)) for _ in range(meta["steps"])]
    return x, xs


def bf16_exact(t):
    """The full-width fixtures' inputs are bf16-representable fp32 values (make_golden.py bf16_exact): reference and HIP path start from the same numbers."""
    return t.to(torch.bfloat16).float()


def seg_wide_inputs(meta):
    g = torch.Generator().manual_seed(meta["pseed"])
    pred = bf16_exact(torch.randn(2, 6, meta["d_model"], generator=g))
    feats = [bf16_exact(torch.randn(2, 256, meta["vit_dim"], generator=g)) for _ in range(2)]
    return pred, feats


def projectors_wide_inputs(meta):
    vf = bf16_exact(torch.randn(*meta["vshape"], generator=torch.Generator().manual_seed(meta["vseed"])))
    af = bf16_exact(torch.randn(*meta["ashape"], generator=torch.Generator().manual_seed(meta["aseed"])))
    return vf, af


def clip_wide_video(meta):
    return bf16_exact(synth.synth_video(meta["t_v"], seed=meta["seed"], clip=meta["clip"])[None])


def beats_wide_audio(meta, L):
    return bf16_exact(synth.synth_audio(meta["t_a"][str(L)], L, seed=meta["seed"], clip=meta["clips"][str(L)]))


# ------------------------------------------------------------------ tiny tokenizer for the harness tests / golden
HARNESS_WORDS = ("this is a an video audio image please answer question describe the events and time range that occurred in "
                 "determine occur based on visual information as well start end of these output location coordinates sounding "
                 "object segment out makes sound how many instruments are playing dog you helpful assistant none . , : ? \n "
                 "<<SYS>> <</SYS>> [INST] [/INST]").split(" ")
LLAMA2_CHAT_TEMPLATE = (
    "{% if messages[0]['role'] == 'system' %}{% set loop_messages = messages[1:] %}{% set system_message = messages[0]['content'] %}"
    "{% else %}{% set loop_messages = messages %}{% set system_message = false %}{% endif %}"
    "{% for message in loop_messages %}{% if loop.index0 == 0 and system_message != false %}"
    "{% set content = '<<SYS>>\\n' + system_message + '\\n<</SYS>>\\n\\n' + message['content'] %}{% else %}{% set content = message['content'] %}{% endif %}"
    "{% if message['role'] == 'user' %}{{ bos_token + '[INST] ' + content.strip() + ' [/INST]' }}"
    "{% elif message['role'] == 'assistant' %}{{ ' ' + content.strip() + ' ' + eos_token }}{% endif %}{% endfor %}")


def tiny_tokenizer(chat_template: bool = True, pad_to: int = 0):
    """A word-level PreTrainedTokenizerFast built in memory (no files): lower-cased words split on whitespace (newline kept as
    a token) and punctuation, <s> / </s> / <unk>, a Llama-2 style chat template, and room for initialize_MM_tokenizer's
    added tokens.  Used by the harness golden (tests/golden/make_golden.py harness) and by tests/test_harness.py."""
    from tokenizers import Regex, Tokenizer, models, normalizers, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    vocab = {"<unk>": 0, "<s>": 1, "</s>": 2}
    for w in HARNESS_WORDS:
        w = w.lower()
        if w and w not in vocab:
            vocab[w] = len(vocab)
    while len(vocab) < pad_to:                              # filler words: len(tokenizer) == the model's base vocabulary
        vocab[f"w{len(vocab)}"] = len(vocab)
    tk = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tk.normalizer = normalizers.Lowercase()
    tk.pre_tokenizer = pre_tokenizers.Split(Regex(r"\n|[^\s\w<>/\[\]]|[<\[][^\s<>\[\]]*[>\]]|[\w]+"), behavior="removed", invert=True)
    tok = PreTrainedTokenizerFast(tokenizer_object=tk, bos_token="<s>", eos_token="</s>", unk_token="<unk>", pad_token="<unk>")
    if chat_template:
        tok.chat_template = LLAMA2_CHAT_TEMPLATE
    return tok


MM_SPECIAL = ['<image>', '<image_start>', '<image_end>', '<video>', '<video_start>', '<video_end>', '<audio>', '<audio_start>',
              '<audio_end>', '<mask>', '<mask_start>', '<mask_end>'] + [f'<mask_{i}>' for i in range(6)]
