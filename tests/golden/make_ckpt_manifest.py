#!/usr/bin/env python3
"""Checkpoint-SHAPE manifest of the reference at FULL size: every state-dict key and shape the reference's classes expose when they are
built the way scripts/quick_start.py:505-529 builds them (Llama-2-7B-chat + hyper-LoRA on every projection, CLIP ViT-L/14, BEATs
iter3+, both Q-Former projectors, SegModule, 32 017-row embeddings after initialize_MM_tokenizer), plus the key sets of the files it
loads: `finetune_weights.bin` of the hyper-LoRA stage (`--save_modules vl_projector,al_projector,lora`, scripts/finetune/
finetune_hyperlora.sh:50 -> scripts/pretrain/trainer.py:183-197 -> utils/deepspeed_utils.py:56-59), of the AVS stage
(`seg_module,embed_tokens,lm_head`, finetune_hyper_lora_avs.sh:52), the BEATs checkpoint ({'cfg', 'model'},
models/multimodal_encoder.py:157-161) and the HF CLIP vision tower.  Runs HERE (build container) with the reference imported from
/root/reference on the META device - no weights exist or are needed - and writes tests/golden/ckpt_manifest.npz (names + shapes: data,
no reference code).  The parity tests then require crab_amd's full-size modules to expose exactly these keys and shapes
(tests/test_ckpt_shapes.py: on the meta device here, and through a real load_state_dict on the GPU box).

    python tests/golden/make_ckpt_manifest.py
"""
import json
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import ref_shims  # noqa: E402

# BEATs_iter3_plus_AS2M_finetuned_on_AS2M_cpt2.pt: the upstream-published iter3+ configuration (SURVEY.md 8c caveat (ii)); the
# fine-tuned checkpoint additionally carries the 527-class predictor head, which the path never evaluates
BEATS_ITER3_PLUS = dict(input_patch_size=16, embed_dim=512, conv_bias=False, encoder_layers=12, encoder_embed_dim=768, encoder_ffn_embed_dim=3072,
                        encoder_attention_heads=12, activation_fn="gelu", layer_wise_gradient_decay_ratio=0.6, layer_norm_first=False, deep_norm=True,
                        dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, encoder_layerdrop=0.0, dropout_input=0.0, conv_pos=128,
                        conv_pos_groups=16, relative_position_embedding=True, num_buckets=320, max_distance=800, gru_rel_pos=True,
                        finetuned_model=True, predictor_dropout=0.0, predictor_class=527)
LLAMA2_7B = dict(vocab_size=32000, hidden_size=4096, intermediate_size=11008, num_hidden_layers=32, num_attention_heads=32, num_key_value_heads=32,
                 rms_norm_eps=1e-5, rope_theta=10000.0)
QWEN2_7B = dict(vocab_size=152064, hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28, num_key_value_heads=4,
                rms_norm_eps=1e-6, rope_theta=1000000.0)
CLIP_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16, image_size=224, patch_size=14,
                layer_norm_eps=1e-5, projection_dim=768, hidden_act="quick_gelu")


class _Tok:
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n

    def add_tokens(self, toks, special_tokens=False):
        self.n += len(toks)
        return len(toks)


def shapes(sd):
    return {k: list(v.shape) for k, v in sd.items()}


def build(me, qwen: bool, beats):
    from peft_hyper import LoraConfig, get_peft_model
    from transformers import CLIPVisionConfig, CLIPVisionModel
    if qwen:
        from models.unified_qwen import UnifiedForCausalLM
        from transformers import Qwen2Config
        cfg = Qwen2Config(**QWEN2_7B, max_position_embeddings=2048, tie_word_embeddings=False, use_sliding_window=False)
    else:
        from models.unified_llama import UnifiedForCausalLM
        from transformers import LlamaConfig
        cfg = LlamaConfig(**LLAMA2_7B, max_position_embeddings=2048, tie_word_embeddings=False, attention_bias=False, pretraining_tp=1)
    cfg._attn_implementation = "eager"
    base = UnifiedForCausalLM(cfg)
    peft_config = LoraConfig(task_type="CAUSAL_LM", target_modules="q_proj,k_proj,v_proj,o_proj,gate_proj,down_proj,up_proj".split(','),
                             inference_mode=False, r=8, lora_alpha=16, lora_dropout=0.05, lora_nums=3)
    model = get_peft_model(base, peft_config)
    inner = model.get_model()
    D = cfg.hidden_size
    # init_multimodal_modules (unified_arch.py:30-113) with its default arguments; the two encoders are attached by hand because their
    # constructors read checkpoint files from a cluster path (multimodal_encoder.py:46-47, 157)
    ccfg = CLIPVisionConfig(**CLIP_L14)

    class VE(me.VisualEncoder):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.select_layer_list, self.select_feature = [14, 22, 23], 'patch'
            self.vision_tower = CLIPVisionModel(ccfg)

    class AE(me.AudioEncoder):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.audio_encoder = beats                  # built on the CPU (its init_bert_params copies through .cpu(): no meta support), 90 M parameters
    inner.visual_encoder = VE()
    inner.vl_projector = me.VLProjector(hidden_size=1024, d_model=D, depth=2, image_token_nums=256, num_query_token=32, num_hidden_layers=2)
    inner.audio_encoder = AE()
    inner.al_projector = me.ALProjector(hidden_size=768, d_model=D, depth=2, num_query_token=32, num_hidden_layers=2)
    inner.seg_module = me.SegModule(d_model=D, prompt_embed_dim=256, image_scale_nums=2, token_nums_per_scale=3, mask_decoder_transformer_depth=2,
                                    vit_image_embedding_dim=1024, avs_query_num=300, num_classes=1, query_generator_num_layers=2, image_size=224,
                                    patch_size=14, image_embedding_size=16, dice_loss_weight=0.5, bce_loss_weight=2.0)
    um = model.base_model.model
    orig = um.resize_token_embeddings
    # transformers 5.x initialises the added rows from the mean / covariance of the old ones (needs data); 4.37.2, the pinned version, does
    # not: plain resize, which is also all the meta device can do
    um.resize_token_embeddings = lambda n, *a, **k: orig(n, mean_resizing=False)
    um.initialize_MM_tokenizer(_Tok(cfg.vocab_size), mask_token_nums=6, use_vqgan=False)
    return model


def main():
    ref_shims.install()
    from transformers.models.bert.configuration_bert import BertConfig
    me = ref_shims.patch_bert_config(lambda: BertConfig())             # bert-base-uncased defaults (multimodal_encoder.py:90,105,192,206)
    out = {}
    from models.beats.BEATs import BEATs, BEATsConfig
    beats = BEATs(BEATsConfig(BEATS_ITER3_PLUS))
    with torch.device("meta"):
        for name, qwen in (("llama", False), ("qwen", True)):
            model = build(me, qwen, beats)
            named = dict(model.named_parameters())
            sec = {"state_dict": shapes(model.state_dict()),
                   # what the trainer writes: named_parameters() filtered by substring (utils/deepspeed_utils.py:56-59)
                   "finetune_hyperlora": shapes({k: v for k, v in named.items() if any(m in k for m in ("vl_projector", "al_projector", "lora"))}),
                   "finetune_avs": shapes({k: v for k, v in named.items() if any(m in k for m in ("seg_module", "embed_tokens", "lm_head"))}),
                   "buffers": sorted(k for k, _ in model.named_buffers())}
            if not qwen:
                inner = model.get_model()
                sec["beats_ckpt_model"] = shapes(inner.audio_encoder.audio_encoder.state_dict())
                sec["clip_hf"] = shapes(inner.visual_encoder.vision_tower.state_dict())
            out[name] = sec
            print(name, {k: len(v) for k, v in sec.items()})
    out["beats_cfg"] = BEATS_ITER3_PLUS
    import transformers
    out["generated_with"] = {"torch": torch.__version__, "transformers": transformers.__version__,
                             "note": "reference classes imported from /root/reference on the meta device; requirements.txt pins transformers 4.37.2"}
    blob = json.dumps(out, separators=(",", ":")).encode()
    np.savez_compressed(os.path.join(HERE, "ckpt_manifest.npz"), manifest=np.frombuffer(blob, dtype=np.uint8))
    print("wrote ckpt_manifest.npz", len(blob), "bytes of json")


if __name__ == "__main__":
    main()
