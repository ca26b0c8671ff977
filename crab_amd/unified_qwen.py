"""Mirror of reference `models/unified_qwen.py` (UnifiedConfig / UnifiedModel / UnifiedForCausalLM over Qwen2).

The reference file is stale against `unified_arch.py` (it passes `batch_attenion_mask=` / `batch_question=` that
`prepare_multimodal_inputs` does not accept, unified_qwen.py:67-75,133-138 -> TypeError as shipped).
As SURVEY.md 8a-12 prescribes, this module keeps the class names / module boundary and gives the Qwen2 decoder
(GQA 28/4, q/k/v bias, models/qwen/modeling_qwen2.py:202-317) the working `generate()` semantics of
`unified_llama.py`.  All kernels are shared with the Llama path: GQA is a head-group index in the attention
kernels, the q/k/v bias rides in the fused QKV GEMM epilogue.
"""
from __future__ import annotations

from dataclasses import dataclass

from .decoder import DecoderConfig
from .unified_llama import UnifiedForCausalLM as _LlamaUnified
from .unified_llama import UnifiedModel as _LlamaUnifiedModel


@dataclass
class UnifiedConfig(DecoderConfig):
    """Qwen2-7B-Instruct defaults (external checkpoint config)."""
    hidden_size: int = 3584
    intermediate_size: int = 18944
    num_hidden_layers: int = 28
    num_attention_heads: int = 28
    num_key_value_heads: int = 4
    vocab_size: int = 152064
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    attention_bias: bool = True
    model_type: str = "unified_llm"


class UnifiedModel(_LlamaUnifiedModel):
    config_class = UnifiedConfig


class UnifiedForCausalLM(_LlamaUnified):
    config_class = UnifiedConfig

    def __init__(self, config: DecoderConfig, device="cuda", **kwargs):
        if not config.attention_bias:
            raise ValueError("Qwen2 has bias on q/k/v (modeling_qwen2.py:234-236): set attention_bias=True")
        super().__init__(config, device=device, **kwargs)
