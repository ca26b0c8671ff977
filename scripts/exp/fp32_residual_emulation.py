#!/usr/bin/env python3
"""How close can a bf16-STORAGE execution get to the fp32 reference (north_star: "logits within 1e-3")?  CPU only, oracle only.

For a decoder (the tiny golden fixture, or a synthetic full-width one) three executions of oracle/crab_oracle.py on the same
bf16-rounded weights and inputs:
  fp32      : no intermediate rounding (the reference arithmetic)
  bf16      : every storage point the HIP path has (projection outputs, q/k after RoPE, P, attention output, SwiGLU product,
              the residual stream, the norm outputs) rounded to bf16, all inner arithmetic fp32 - an EXACT bf16-storage execution
  bf16+res32: the same, but the residual stream kept in fp32
and prints max |logit - logit_fp32| / max |logit_fp32| of the last row.

    python scripts/exp/fp32_residual_emulation.py tiny
    python scripts/exp/fp32_residual_emulation.py full --layers 32 --seq 702      # ~2 min on 128+ host threads
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import crab_oracle as O  # noqa: E402

BF = torch.bfloat16


def run(W, cfg, emb):
    out = {}
    ref = O.decoder_forward(emb, W, cfg, last_only=True)[0][:, -1]
    O.EMULATE_FP32_RESIDUAL = False
    a = O.decoder_forward(emb, W, cfg, last_only=True, emulate=BF)[0][:, -1]
    O.EMULATE_FP32_RESIDUAL = True
    b = O.decoder_forward(emb, W, cfg, last_only=True, emulate=BF)[0][:, -1]
    sc = ref.abs().max().item()
    out["scale"] = sc
    out["bf16_storage_rel"] = (a - ref).abs().max().item() / sc
    out["bf16_storage_fp32_residual_rel"] = (b - ref).abs().max().item() / sc
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", choices=["tiny", "full"])
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--seq", type=int, default=702)
    a = ap.parse_args()
    if a.which == "tiny":
        from tests.util import load_fixture, weights_from_table
        for name in ("full_tiny_llama", "full_tiny_qwen"):
            meta, A = load_fixture(name)
            W = {k: v.to(BF).float() for k, v in O.strip_peft_prefix(weights_from_table(meta)).items()}
            cfg = O.DecoderConfig(**meta["dec"])
            print(name, run(W, cfg, A["embeds_bs1"].to(BF).float()))
        return
    # synthetic full-width Llama-2-7B-size decoder, conditioned like crab_amd.build_model.randomize_(conditioned=True)
    cfg = O.DecoderConfig(num_hidden_layers=a.layers)
    g = torch.Generator().manual_seed(42)
    D, I, V = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size

    def rnd(*s, std=0.02):
        return (torch.randn(*s, generator=g) * std).to(BF).float()

    W = {"model.embed_tokens.weight": rnd(V, D, std=1.0), "lm_head.weight": rnd(V, D), "model.norm.weight": torch.ones(D)}
    for i in range(a.layers):
        q = f"model.layers.{i}"
        W[q + ".input_layernorm.weight"] = torch.ones(D)
        W[q + ".post_attention_layernorm.weight"] = torch.ones(D)
        for n, (o, k) in {"self_attn.q_proj": (D, D), "self_attn.k_proj": (D, D), "self_attn.v_proj": (D, D), "self_attn.o_proj": (D, D),
                          "mlp.gate_proj": (I, D), "mlp.up_proj": (I, D), "mlp.down_proj": (D, I)}.items():
            small = 0.1 if n in ("self_attn.o_proj", "mlp.down_proj") else 1.0
            W[f"{q}.{n}.weight"] = rnd(o, k, std=0.02 * small)
            W[f"{q}.{n}.lora_route.weight"], W[f"{q}.{n}.lora_A.weight"] = rnd(3, k), rnd(8, k)
            for j in range(3):
                W[f"{q}.{n}.lora_B{j}.weight"] = rnd(o, 8, std=0.002)
    emb = torch.randn(1, a.seq, D, generator=g).to(BF).float()
    print(f"full-width, {a.layers} layers, S={a.seq}:", run(W, cfg, emb))


if __name__ == "__main__":
    main()
