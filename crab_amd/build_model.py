"""Construction helpers for full-size synthetic models (no checkpoints exist offline).

`build_crab(...)` follows the order of scripts/quick_start.py:465-529: UnifiedForCausalLM -> get_peft_model ->
init_multimodal_modules -> initialize_MM_tokenizer, then fills every parameter in place with N(0, 0.02^2)
(norm weights 1) using the device RNG -- synthetic data generation, not model arithmetic.
"""
from __future__ import annotations

from typing import Optional

import torch

from .peft_hyper import LoraConfig, get_peft_model

BF16 = torch.bfloat16


class CountingTokenizer:
    """Stands in for the LLM tokenizer where only len()/add_tokens() are needed (initialize_MM_tokenizer)."""

    def __init__(self, n: int):
        self.n = n
        self.added = []

    def __len__(self):
        return self.n

    def add_tokens(self, toks, special_tokens=False):
        self.added += list(toks)
        self.n += len(toks)
        return len(toks)


@torch.no_grad()
def randomize_(model: torch.nn.Module, seed: int = 42, std: float = 0.02, conditioned: bool = False):
    """In-place synthetic weights: matrices ~ N(0, std^2), 1-D norm weights = 1, biases ~ N(0, std^2).
    conditioned=True keeps the decoder's sub-layer outputs small against the residual stream (embeddings ~ N(0,1),
    o_proj / down_proj / lora_B x0.1) like a trained pre-LN decoder: used by the full-size parity-property tests, where
    a fully random 32-layer decoder would amplify bf16 rounding chaotically."""
    dev = next(model.parameters()).device
    g = torch.Generator(device=dev).manual_seed(seed)
    for name, p in model.named_parameters():
        leaf = name.rsplit(".", 1)[-1]
        parent = name.rsplit(".", 1)[0].rsplit(".", 1)[-1].lower()
        if p.dim() == 1 and leaf == "weight" and ("norm" in parent or "ln" in parent):
            p.fill_(1.0)
        elif leaf in ("grep_a", "weight_g"):
            p.fill_(1.0)
        else:
            sd_ = std
            if conditioned:
                if "embed_tokens" in name:
                    sd_ = 1.0
                elif ".o_proj.weight" in name or ".down_proj.weight" in name or ".lora_B" in name:
                    sd_ = std * 0.1
            tmp = torch.empty(p.shape, device=dev, dtype=torch.float32).normal_(0.0, sd_, generator=g)
            p.copy_(tmp)
    return model


def build_crab(llm: str = "llama", device="cuda", num_hidden_layers: Optional[int] = None, seed: int = 42,
               visual: bool = True, audio: bool = True, randomize: bool = True, conditioned: bool = False, segment: bool = False,
               vqgan: bool = False):
    """Full-size Crab (Llama-2-7B or Qwen2-7B decoder + CLIP ViT-L/14 + BEATs iter3+ + Q-Former projectors; segment=True adds
    the SegModule of the AVS tasks, vqgan=True the taming f16/16384 MaskEncoder, as scripts/quick_start.py:505-529 would)."""
    if llm == "llama":
        from .unified_llama import UnifiedConfig, UnifiedForCausalLM
        cfg = UnifiedConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32,
                            num_key_value_heads=32, vocab_size=32000, rms_norm_eps=1e-5, rope_theta=10000.0,
                            pad_token_id=2, eos_token_id=2)
    elif llm == "qwen":
        from .unified_qwen import UnifiedConfig, UnifiedForCausalLM
        cfg = UnifiedConfig(pad_token_id=151643, eos_token_id=151645)
    else:
        raise ValueError(llm)
    if num_hidden_layers is not None:
        cfg.num_hidden_layers = num_hidden_layers
    base_vocab = cfg.vocab_size
    model = get_peft_model(UnifiedForCausalLM(cfg, device=device), LoraConfig())
    model.get_model().pad_token_id = cfg.pad_token_id
    model.get_model().init_multimodal_modules(d_model=cfg.hidden_size, visual_branch=visual, audio_branch=audio,
                                              select_layer_list=[14, 22, 23], segment_branch=segment, use_vqgan=vqgan)
    model.initialize_MM_tokenizer(CountingTokenizer(base_vocab), mask_token_nums=6)
    model.base_vocab = base_vocab
    if randomize:
        randomize_(model, seed, conditioned=conditioned)
    model.eval()
    return model
