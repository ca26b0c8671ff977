"""CPU ORACLE for the Crab inference hot path -- TEST INFRASTRUCTURE ONLY.

A plain PyTorch fp32 restatement of the reference's forward arithmetic for the path named by
BASELINE.json (BEATs -> CLIP ViT -> Q-Former projectors -> hyper-LoRA Llama/Qwen2 decoder ->
greedy decode), written from SURVEY.md Appendix B and the cited reference lines.  Nothing under
crab_amd/ may import this file: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg use it, and only as the checker / CPU baseline, never as the thing measured or shipped.

Pinning: the reference ships no tests and no golden vectors for this path ("parity unpinned" by
the reference itself).  The pins are therefore the fixtures under tests/golden/*.npz,
produced by tests/golden/make_golden.py, which imports the reference from /root/reference in the build
container, loads crab_amd.synth weights into it and records its outputs.  tests/test_oracle_golden.py
checks every function below against those fixtures (fp32, tolerance 2e-4 abs on O(1) values; the
reference runs on transformers 5.15 here instead of the pinned 4.37.2, see SURVEY.md 8c caveats).

All weights are addressed by the reference's own state-dict key names (minus PEFT's
`base_model.model.` prefix), so a `finetune_weights.bin`-style dict drives the oracle directly.

Device: the functions are plain PyTorch and run wherever their inputs live.  The fixtures, bench.py's cpu_baseline and most tests run them
on the HOST (fp32 eager).  The decoder functions create their index tensors on the input's device, so the three full-depth tests of
tests/test_fullsize_gpu.py may hand in weights and embeddings that sit on the GPU (torch's own fp32 kernels, TF32 off): the same fp32
arithmetic up to summation order (tests/test_fullsize_gpu.py::test_oracle_on_the_gpu_equals_the_oracle_on_the_host pins the two to 2e-5)
in seconds instead of minutes of host time that varied 2.5x between GPU boxes (profiles/README.md r05).

`emulate` argument: when a torch dtype (bf16) is given, activations are rounded to that dtype at the
same points where the HIP path stores bf16 tensors (GEMM outputs, norm outputs, attention output),
with all inner arithmetic in fp32.  emulate=None is the exact fp32 path of the reference as shipped
(scripts/quick_start.sh:42-44 pass --bf16 False).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
WDict = Dict[str, Tensor]


class OperandRounding:
    """emulate=OPERANDS: the bf16-OPERAND FLOOR of any MFMA implementation of this path.  Only what a matrix instruction consumes is rounded,
    once, at the point of consumption: the weights, the input rows of every linear layer / convolution (incl. the routed rank-r activations
    `u` of the hyper-LoRA update) and the attention operands q (after RoPE), k (after RoPE) and v.  Everything else - residual streams, norm
    statistics and outputs as residual terms, pre-rotation q / k, softmax probabilities, GEMM outputs that are not themselves operands -
    stays fp32.  No storage format can do better with bf16 matrix operands, so max|floor - fp32| is the part of a measured HIP error that is
    irreducible; the distance between the floor and the storage emulation (emulate=torch.bfloat16) is what removable storage points cost."""

    def __init__(self, dtype=torch.bfloat16):
        self.dtype = dtype


OPERANDS = OperandRounding()


# Storage points (by tag) that the storage emulation treats as fp32: the ablation switch of scripts/parity_floor.py (which storage point costs
# what) and the place a storage point REMOVED from the HIP path is recorded.  Tags: "ln" (LayerNorm outputs of the post-LN encoders, which are
# residual terms as well as operands), "p" (softmax probabilities), "embed" (patch / position embedding sums), "kept" (CLIP hidden states
# handed to the projector), "out" (projector outputs).
STORAGE_FP32: set = set()


def _r(x: Tensor, emulate, tag: Optional[str] = None) -> Tensor:
    """A STORAGE point of the HIP path: round-trip through the emulated storage dtype (identity for emulate=None and for the operand floor)."""
    if emulate is None or isinstance(emulate, OperandRounding) or (tag is not None and tag in STORAGE_FP32):
        return x
    return x.to(emulate).to(torch.float32)


def _op(x: Tensor, emulate) -> Tensor:
    """A matrix OPERAND at its point of consumption: rounded to bf16 in both emulation modes (idempotent in the storage emulation wherever the
    producer already rounded it where the HIP path stores it; emulate=None is exact)."""
    if emulate is None:
        return x
    return x.to(emulate.dtype if isinstance(emulate, OperandRounding) else emulate).to(torch.float32)


def _ro(x: Tensor, emulate) -> Tensor:
    """Both a storage point and an operand (rotated q / k, the routed activations u): rounded once in either emulation mode."""
    return _op(_r(x, emulate), emulate)


# With emulate=bfloat16: the RESIDUAL STREAM (decoder x, CLIP tower x) stays fp32 and RMSNorm's x_hat is not rounded before the
# weight multiply, every other storage point still rounds - the storage points of the HIP path since r04 (crab_amd.ops.RESIDUAL_FP32).
# False = the all-bf16 storage of r01-r03 (scripts/exp/fp32_residual_emulation.py compares the two).
EMULATE_FP32_RESIDUAL = True


class residual_storage:
    """`with residual_storage(fp32=False):` emulate the all-bf16 storage of r01-r03 inside the block (tests of the bf16-residual form of the kernels)."""

    def __init__(self, fp32: bool):
        self.fp32 = bool(fp32)

    def __enter__(self):
        global EMULATE_FP32_RESIDUAL
        self.prev, EMULATE_FP32_RESIDUAL = EMULATE_FP32_RESIDUAL, self.fp32

    def __exit__(self, *a):
        global EMULATE_FP32_RESIDUAL
        EMULATE_FP32_RESIDUAL = self.prev


def _rres(x: Tensor, emulate) -> Tensor:
    return x if EMULATE_FP32_RESIDUAL else _r(x, emulate)


def strip_peft_prefix(sd: WDict) -> WDict:
    """finetune_weights.bin keys carry PEFT's `base_model.model.` prefix."""
    out = {}
    for k, v in sd.items():
        out[k[len("base_model.model."):] if k.startswith("base_model.model.") else k] = v
    return out


# =====================================================================================
# B.1 hyper-LoRA Linear                       reference peft_hyper/tuners/lora.py:338-350
# =====================================================================================

def linear(x: Tensor, W: WDict, prefix: str, emulate=None, store: bool = True) -> Tensor:
    """store=False: the output is not a storage point of the HIP path (consumed inside a fused epilogue in fp32)."""
    w = W[prefix + ".weight"]
    b = W.get(prefix + ".bias")
    y = F.linear(_op(x, emulate), _op(w, emulate), b)
    return _r(y, emulate) if store else y


def hyperlora_linear(x: Tensor, W: WDict, prefix: str, scaling: float = 2.0, lora_nums: int = 3,
                     emulate=None, store: bool = True) -> Tensor:
    """y = x W^T (+b) + sum_i softmax_fp32(x R^T)_i * (B_i (A x)) * scaling   (lora.py:341-350).

    Falls back to a plain Linear when the prefix carries no lora_A (module not wrapped).
    Emulation: the routed activations u_i = scaling * softmax_i * (A x) are an operand of the K-extended GEMM of the HIP path (bf16 `u`,
    crab_hyperlora_mix), so they are rounded once in both emulation modes; in fp32 the order of the scalar factors is the reference's."""
    w = W[prefix + ".weight"]
    b = W.get(prefix + ".bias")
    xo = _op(x, emulate)
    y = F.linear(xo, _op(w, emulate), b)
    if (prefix + ".lora_A.weight") not in W:
        return _r(y, emulate) if store else y
    route = torch.softmax(F.linear(xo, _op(W[prefix + ".lora_route.weight"], emulate)).float(), dim=-1)   # lora.py:346
    h = F.linear(xo, _op(W[prefix + ".lora_A.weight"], emulate))                                          # lora.py:349
    for i in range(lora_nums):
        if emulate is None:
            y = y + route[..., i:i + 1] * F.linear(h, W[prefix + f".lora_B{i}.weight"]) * scaling
        else:
            y = y + F.linear(_ro(route[..., i:i + 1] * h * scaling, emulate), _op(W[prefix + f".lora_B{i}.weight"], emulate))
    return _r(y, emulate) if store else y


# =====================================================================================
# B.2 Llama / Qwen2 decoder        reference models/modeling_llama.py, models/qwen/modeling_qwen2.py
# =====================================================================================

@dataclass
class DecoderConfig:
    hidden_size: int = 4096
    intermediate_size: int = 11008
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 32
    vocab_size: int = 32017
    rms_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    lora_r: int = 8
    lora_alpha: int = 16
    lora_nums: int = 3

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def scaling(self) -> float:
        return self.lora_alpha / self.lora_r

    @staticmethod
    def llama2_7b() -> "DecoderConfig":
        return DecoderConfig()

    @staticmethod
    def qwen2_7b() -> "DecoderConfig":
        return DecoderConfig(hidden_size=3584, intermediate_size=18944, num_hidden_layers=28,
                             num_attention_heads=28, num_key_value_heads=4, vocab_size=152064 + 17,
                             rms_norm_eps=1e-6, rope_theta=1e6)


def rmsnorm(x: Tensor, w: Tensor, eps: float, emulate=None) -> Tensor:
    """modeling_llama.py:112-117: fp32 variance, x_hat cast to input dtype, then * weight."""
    x32 = x.float()
    xh = x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + eps)
    if EMULATE_FP32_RESIDUAL:                 # fp32 row in: bf16(w * x_hat), one rounding (crab_rmsnorm_f32)
        return _r(w.float() * xh, emulate)
    return _r(w.float() * _r(xh, emulate), emulate)


def rope_cos_sin(positions: Tensor, head_dim: int, theta: float) -> Tuple[Tensor, Tensor]:
    """modeling_llama.py:130-156: inv_freq_i = theta^(-2i/d); emb = cat(freqs, freqs)."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32, device=positions.device) / head_dim))
    fr = positions.float()[..., None] * inv            # [..., d/2]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def _rot_half(x: Tensor) -> Tensor:
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)   # modeling_llama.py:204-208


def apply_rope(q: Tensor, k: Tensor, cos: Tensor, sin: Tensor) -> Tuple[Tensor, Tensor]:
    """q,k: [b,h,s,d]; cos,sin: [b,s,d] (modeling_llama.py:211-236)."""
    c, s = cos[:, None], sin[:, None]
    return q * c + _rot_half(q) * s, k * c + _rot_half(k) * s


@dataclass
class KVCache:
    """DynamicCache equivalent: per-layer [b,h_kv,t,d] tensors appended after RoPE (:408-412)."""
    k: List[Optional[Tensor]] = field(default_factory=list)
    v: List[Optional[Tensor]] = field(default_factory=list)

    def length(self) -> int:
        return 0 if not self.k or self.k[0] is None else self.k[0].shape[2]


def decoder_layer(x: Tensor, W: WDict, i: int, cfg: DecoderConfig, cache: KVCache, positions: Tensor,
                  emulate=None, key_mask: Optional[Tensor] = None) -> Tensor:
    """One Llama/Qwen2 layer (modeling_llama.py:805-827; attention :394-452; MLP :269).
    x [b,s,D]; positions [b,s] absolute position ids; key_mask [b,t] (1 = attend) is the 2-D attention_mask over ALL keys
    (cached + new) that forward() passes on (unified_llama.py:149-160): combined with the causal mask as an additive
    finfo.min term (modeling_llama.py:420-428).  A query row whose keys are all masked (a left-pad row) gets an undefined
    (implementation-dependent) output that no valid row ever reads; callers compare valid rows only."""
    p = f"model.layers.{i}"
    b, s, D = x.shape
    H, Hk, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    sc, ln = cfg.scaling, cfg.lora_nums

    h = rmsnorm(x, W[p + ".input_layernorm.weight"], cfg.rms_norm_eps, emulate)
    q = hyperlora_linear(h, W, p + ".self_attn.q_proj", sc, ln, emulate).view(b, s, H, d).transpose(1, 2)
    k = hyperlora_linear(h, W, p + ".self_attn.k_proj", sc, ln, emulate).view(b, s, Hk, d).transpose(1, 2)
    v = hyperlora_linear(h, W, p + ".self_attn.v_proj", sc, ln, emulate).view(b, s, Hk, d).transpose(1, 2)
    cos, sin = rope_cos_sin(positions, d, cfg.rope_theta)
    q, k = apply_rope(q, k, cos, sin)
    q, k, v = _ro(q, emulate), _ro(k, emulate), _op(v, emulate)             # the attention operands (k / v as the cache holds them)
    while len(cache.k) <= i:
        cache.k.append(None)
        cache.v.append(None)
    if cache.k[i] is None:
        cache.k[i], cache.v[i] = k, v
    else:
        cache.k[i] = torch.cat([cache.k[i], k], dim=2)
        cache.v[i] = torch.cat([cache.v[i], v], dim=2)
    kk, vv = cache.k[i], cache.v[i]
    t = kk.shape[2]
    if Hk != H:                                           # repeat_kv, modeling_llama.py:274-283
        g = H // Hk
        kk = kk[:, :, None].expand(b, Hk, g, t, d).reshape(b, H, t, d)
        vv = vv[:, :, None].expand(b, Hk, g, t, d).reshape(b, H, t, d)
    a = torch.matmul(q, kk.transpose(2, 3)) / math.sqrt(d)                    # :417
    # causal mask: query row r (absolute index t-s+r) sees keys 0..t-s+r       (:420-428)
    qi = torch.arange(t - s, t, device=x.device)[:, None]
    kj = torch.arange(t, device=x.device)[None, :]
    a = a.masked_fill((kj > qi)[None, None], torch.finfo(torch.float32).min)
    if key_mask is not None:
        a = a.masked_fill((key_mask[:, None, None, :t] == 0), torch.finfo(torch.float32).min)
    pr = torch.softmax(a.float(), dim=-1)                                      # :431 fp32 softmax
    o = torch.matmul(_r(pr, emulate), vv).transpose(1, 2).reshape(b, s, H * d)
    o = _r(o, emulate)
    x = _rres(x + hyperlora_linear(o, W, p + ".self_attn.o_proj", sc, ln, emulate, store=False), emulate)

    h = rmsnorm(x, W[p + ".post_attention_layernorm.weight"], cfg.rms_norm_eps, emulate)
    g_ = hyperlora_linear(h, W, p + ".mlp.gate_proj", sc, ln, emulate, store=False)
    u_ = hyperlora_linear(h, W, p + ".mlp.up_proj", sc, ln, emulate, store=False)
    m = _r(F.silu(g_) * u_, emulate)                                           # :269
    x = _rres(x + hyperlora_linear(m, W, p + ".mlp.down_proj", sc, ln, emulate, store=False), emulate)
    return x


def decoder_forward(embeds: Tensor, W: WDict, cfg: DecoderConfig, cache: Optional[KVCache] = None,
                    positions: Optional[Tensor] = None, last_only: bool = False, emulate=None,
                    attention_mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, KVCache]:
    """LlamaModel.forward + lm_head (modeling_llama.py:989-1124, 1169-1287).
    Returns (logits fp32, post-final-norm hidden, cache).  `last_only` computes lm_head on the last
    row only (output-identical for generate())."""
    b, s, _ = embeds.shape
    cache = cache if cache is not None else KVCache()
    past = cache.length()
    if positions is None:
        positions = torch.arange(past, past + s, device=embeds.device)[None].expand(b, s)
    x = _r(embeds.float(), emulate)
    for i in range(cfg.num_hidden_layers):
        x = decoder_layer(x, W, i, cfg, cache, positions, emulate, key_mask=attention_mask)
    hn = rmsnorm(x, W["model.norm.weight"], cfg.rms_norm_eps, emulate)
    hh = hn[:, -1:] if last_only else hn
    logits = F.linear(_op(hh, emulate), _op(W["lm_head.weight"], emulate)).float()
    return logits, hn, cache


def greedy_generate(embeds: Tensor, W: WDict, cfg: DecoderConfig, max_new_tokens: int,
                    eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None,
                    min_new_tokens: int = 0, emulate=None, return_hidden: bool = False):
    """B.3: HF GenerationMixin greedy loop as driven by unified_llama.py:262-267 with only
    inputs_embeds: positions 0..S-1 (pads attended, no mask forwarded), argmax on fp32 last-row logits,
    finished rows emit pad, returns ONLY new ids.  Also returns per-step last-row logits."""
    b = embeds.shape[0]
    cache = KVCache()
    logits, hn, cache = decoder_forward(embeds, W, cfg, cache, last_only=True, emulate=emulate)
    ids, step_logits, hiddens = [], [], []
    unfinished = torch.ones(b, dtype=torch.bool, device=embeds.device)
    pad = pad_token_id if pad_token_id is not None else (eos_token_id if eos_token_id is not None else 0)
    for step in range(max_new_tokens):
        lg = logits[:, -1].clone()
        step_logits.append(lg)
        hiddens.append(hn[:, -1])
        if eos_token_id is not None and step < min_new_tokens:
            lg[:, eos_token_id] = -float("inf")
        nxt = lg.argmax(-1)
        nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
        ids.append(nxt)
        if eos_token_id is not None:
            unfinished = unfinished & (nxt != eos_token_id)
            if not unfinished.any():
                break
        if step + 1 == max_new_tokens:
            break
        e = W["model.embed_tokens.weight"][nxt][:, None]              # unified_llama.py:125-127
        logits, hn, cache = decoder_forward(e, W, cfg, cache, last_only=True, emulate=emulate)
    out = torch.stack(ids, dim=1)
    sl = torch.stack(step_logits, dim=1)
    if return_hidden:
        return out, sl, torch.stack(hiddens, dim=1)
    return out, sl


def sampling_probs(logits: Tensor, temperature: float = 0.6, top_k: int = 50, top_p: float = 0.9) -> Tensor:
    """The distribution HF's sample mode draws from (third-party: transformers==4.37.2 generation/logits_process.py, applied in the order
    of generation/utils.py:_get_logits_warper - TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper - then softmax; the reference
    reaches it because scripts/quick_start.py:36-43 never passes do_sample and Llama-2-chat's generation_config sets it).
    logits [B, V] fp32 -> probabilities [B, V] (zeros outside the kept set).  Pinned against the installed transformers' own warper
    classes in tests/test_oracle_golden.py."""
    x = logits.float() / temperature
    if top_k and top_k < x.shape[-1]:
        kth = x.topk(top_k, dim=-1).values[..., -1:]
        x = x.masked_fill(x < kth, -float("inf"))
    if top_p < 1.0:
        sl, si = torch.sort(x, descending=False, dim=-1)
        cum = sl.softmax(-1).cumsum(-1)
        rem = cum <= (1 - top_p)
        rem[..., -1:] = False                                            # min_tokens_to_keep = 1
        x = x.masked_fill(rem.scatter(-1, si, rem), -float("inf"))
    return x.softmax(-1)


# =====================================================================================
# B.4 CLIP ViT vision tower (HF CLIPVisionModel; Crab use: multimodal_encoder.py:52-84)
# =====================================================================================

@dataclass
class ClipConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    layer_norm_eps: float = 1e-5
    select_layers: Tuple[int, ...] = (14, 22, 23)

    @property
    def num_patches(self) -> int:
        return (self.image_size // self.patch_size) ** 2


def layernorm(x: Tensor, W: WDict, prefix: str, eps: float, emulate=None, tag: Optional[str] = "ln") -> Tensor:
    return _r(F.layer_norm(x.float(), (x.shape[-1],), W[prefix + ".weight"].float(), W[prefix + ".bias"].float(), eps),
              emulate, tag)


def _mha(q: Tensor, k: Tensor, v: Tensor, heads: int, scale: float, bias: Optional[Tensor] = None,
         emulate=None) -> Tensor:
    """q [b,n,D], k/v [b,m,D] -> [b,n,D]; softmax(q k^T * scale + bias) v."""
    b, n, D = q.shape
    m = k.shape[1]
    d = D // heads
    qh = _op(q, emulate).view(b, n, heads, d).transpose(1, 2)
    kh = _op(k, emulate).view(b, m, heads, d).transpose(1, 2)
    vh = _op(v, emulate).view(b, m, heads, d).transpose(1, 2)
    a = torch.matmul(qh, kh.transpose(2, 3)) * scale
    if bias is not None:
        a = a + bias
    p = _r(torch.softmax(a.float(), dim=-1), emulate, "p")
    o = torch.matmul(p, vh).transpose(1, 2).reshape(b, n, D)
    return _r(o, emulate)


def clip_vision(pixels: Tensor, W: WDict, cfg: ClipConfig, prefix: str = "model.visual_encoder.vision_tower.vision_model",
                emulate=None, max_layer: Optional[int] = None) -> List[Tensor]:
    """pixels [N,3,H,W] -> hidden_states list h_0..h_L, each [N, 1+P, D] (B.4).
    Stops after max(select_layers) layers when max_layer is None (layer 24 / post_layernorm are dead,
    SURVEY.md appendix A.2)."""
    N = pixels.shape[0]
    ps = cfg.patch_size
    w = W[prefix + ".embeddings.patch_embedding.weight"]
    x = F.conv2d(_op(pixels.float(), emulate), _op(w, emulate), None, stride=ps)   # [N,D,g,g], no bias
    x = _r(x.flatten(2).transpose(1, 2), emulate, "embed")               # [N,P,D]
    cls = W[prefix + ".embeddings.class_embedding"].float().expand(N, 1, -1)
    x = torch.cat([cls, x], dim=1) + W[prefix + ".embeddings.position_embedding.weight"].float()[None]
    # (the sum [cls | patches] + positions feeds pre_layrnorm unrounded: clip_embed_ln_kernel, r05)
    h = layernorm(x, W, prefix + ".pre_layrnorm", cfg.layer_norm_eps, emulate, "embed")
    hs = [h]
    H = cfg.num_attention_heads
    d = cfg.hidden_size // H
    L = max_layer if max_layer is not None else max(cfg.select_layers)
    for i in range(L):
        p = f"{prefix}.encoder.layers.{i}"
        a = layernorm(h, W, p + ".layer_norm1", cfg.layer_norm_eps, emulate)
        q = linear(a, W, p + ".self_attn.q_proj", emulate)
        k = linear(a, W, p + ".self_attn.k_proj", emulate)
        v = linear(a, W, p + ".self_attn.v_proj", emulate)
        o = _mha(q, k, v, H, d ** -0.5, None, emulate)
        h = _rres(h + linear(o, W, p + ".self_attn.out_proj", emulate, store=False), emulate)
        a = layernorm(h, W, p + ".layer_norm2", cfg.layer_norm_eps, emulate)
        f1 = linear(a, W, p + ".mlp.fc1", emulate, store=False)
        f1 = _r(f1 * torch.sigmoid(1.702 * f1), emulate)                 # quick_gelu
        h = _rres(h + linear(f1, W, p + ".mlp.fc2", emulate, store=False), emulate)
        hs.append(_r(h, emulate, "kept"))                                # the kept states are bf16 (what the projectors read)
    return hs


def visual_encoder(video: Tensor, W: WDict, cfg: ClipConfig, emulate=None,
                   prefix: str = "model.visual_encoder.vision_tower.vision_model") -> List[Tensor]:
    """VisualEncoder.forward (multimodal_encoder.py:75-84): video [b,t,3,H,W] -> list of [b,t*P,D],
    one per select layer, CLS dropped (feature_select :52-63)."""
    b, t = video.shape[:2]
    hs = clip_vision(video.reshape(b * t, *video.shape[2:]), W, cfg, prefix, emulate)
    out = []
    for lyr in cfg.select_layers:
        f = hs[lyr][:, 1:]
        out.append(f.reshape(b, t * f.shape[1], f.shape[2]))
    return out


# =====================================================================================
# B.5 Q-Former projectors    reference models/Qformer.py, multimodal_encoder.py:119-144,226-262
# =====================================================================================

@dataclass
class QFormerConfig:
    hidden_size: int = 768
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    num_hidden_layers: int = 2
    layer_norm_eps: float = 1e-12
    num_query_token: int = 32


def _gelu(x: Tensor) -> Tensor:
    return F.gelu(x)          # exact erf GELU (ACT2FN['gelu'], nn.GELU())


def qformer(query: Tensor, enc: Tensor, W: WDict, prefix: str, cfg: QFormerConfig, emulate=None) -> Tensor:
    """BertModel.forward with query_embeds only (Qformer.py:806-967): query [B,32,h], enc [B,m,enc_w]."""
    eps = cfg.layer_norm_eps
    H = cfg.num_attention_heads
    d = cfg.hidden_size // H
    z = layernorm(query, W, prefix + ".embeddings.LayerNorm", eps, emulate)          # :105-108
    for l in range(cfg.num_hidden_layers):
        p = f"{prefix}.encoder.layer.{l}"
        # self attention (:171-277), scores / sqrt(d), mask all ones -> additive 0
        q = linear(z, W, p + ".attention.self.query", emulate)
        k = linear(z, W, p + ".attention.self.key", emulate)
        v = linear(z, W, p + ".attention.self.value", emulate)
        c = _mha(q, k, v, H, 1.0 / math.sqrt(d), None, emulate)
        z = layernorm(linear(c, W, p + ".attention.output.dense", emulate, store=False) + z, W, p + ".attention.output.LayerNorm",
                      eps, emulate)                                                    # :287-291
        # cross attention every layer (cross_attention_freq=1)
        q = linear(z, W, p + ".crossattention.self.query", emulate)
        k = linear(enc, W, p + ".crossattention.self.key", emulate)
        v = linear(enc, W, p + ".crossattention.self.value", emulate)
        c = _mha(q, k, v, H, 1.0 / math.sqrt(d), None, emulate)
        z = layernorm(linear(c, W, p + ".crossattention.output.dense", emulate, store=False) + z, W,
                      p + ".crossattention.output.LayerNorm", eps, emulate)
        # query FFN (:483-486)
        f = _r(_gelu(linear(z, W, p + ".intermediate_query.dense", emulate, store=False)), emulate)
        z = layernorm(linear(f, W, p + ".output_query.dense", emulate, store=False) + z, W, p + ".output_query.LayerNorm", eps,
                      emulate)
    return z


def vl_projector(feat: Tensor, W: WDict, cfg: QFormerConfig, image_token_nums: int = 256,
                 prefix: str = "model.vl_projector", emulate=None) -> Tensor:
    """VLProjector.forward (multimodal_encoder.py:119-144): [b,t*n,enc] -> [b,t*32,D]."""
    b, tn, dim = feat.shape
    t = tn // image_token_nums
    x = feat.reshape(b * t, image_token_nums, dim)
    x = layernorm(x, W, prefix + ".visual_ln", 1e-5, emulate)
    qt = W[prefix + ".visual_query_tokens"].float().expand(b * t, -1, -1)
    z = qformer(qt, x, W, prefix + ".visual_Qformer.bert", cfg, emulate)[:, :cfg.num_query_token]
    y = _r(_gelu(linear(z, W, prefix + ".visual_proj.0", emulate, store=False)), emulate)
    y = linear(y, W, prefix + ".visual_proj.2", emulate)
    return y.reshape(b, t * cfg.num_query_token, -1)


def al_projector(feat: Tensor, W: WDict, cfg: QFormerConfig, prefix: str = "model.al_projector",
                 emulate=None) -> Tensor:
    """ALProjector.forward 4-D branch (multimodal_encoder.py:226-244): [b,t,n,768] -> [b,t*32,D]."""
    b, t, n, d = feat.shape
    x = layernorm(feat.reshape(b * t, n, d), W, prefix + ".audio_ln", 1e-5, emulate)
    qt = W[prefix + ".audio_query_tokens"].float().expand(b * t, -1, -1)
    z = qformer(qt, x, W, prefix + ".audio_Qformer.bert", cfg, emulate)[:, :cfg.num_query_token]
    z = z.reshape(b, t * cfg.num_query_token, -1)
    y = _r(_gelu(linear(z, W, prefix + ".audio_proj.0", emulate, store=False)), emulate)
    return linear(y, W, prefix + ".audio_proj.2", emulate)


# =====================================================================================
# B.6 BEATs        reference models/beats/BEATs.py:134-182, backbone.py
# =====================================================================================

@dataclass
class BeatsConfig:
    input_patch_size: int = 16
    embed_dim: int = 512
    encoder_embed_dim: int = 768
    encoder_ffn_embed_dim: int = 3072
    encoder_attention_heads: int = 12
    encoder_layers: int = 12
    conv_pos: int = 128
    conv_pos_groups: int = 16
    num_buckets: int = 320
    max_distance: int = 800
    deep_norm: bool = True
    gru_rel_pos: bool = True
    conv_bias: bool = False
    layer_norm_eps: float = 1e-5


def rel_pos_bucket(qlen: int, klen: int, num_buckets: int, max_distance: int) -> Tensor:
    """backbone.py:392-430 (bidirectional T5 buckets): int64 [qlen,klen]."""
    ctx = torch.arange(qlen, dtype=torch.long)[:, None]
    mem = torch.arange(klen, dtype=torch.long)[None, :]
    rel = mem - ctx
    nb = num_buckets // 2
    out = (rel > 0).to(torch.long) * nb
    rel = rel.abs()
    max_exact = nb // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
                         * (nb - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return out + torch.where(is_small, rel, large)


def beats(fbank: Tensor, W: WDict, cfg: BeatsConfig, prefix: str = "model.audio_encoder.audio_encoder",
          emulate=None) -> Tensor:
    """BEATs.extract_features(feature_only=True) with an all-False padding mask: [B,L,128] -> [B,n,768]."""
    B = fbank.shape[0]
    P = cfg.input_patch_size
    E = cfg.encoder_embed_dim
    H = cfg.encoder_attention_heads
    d = E // H
    x = F.conv2d(_op(fbank.float()[:, None], emulate), _op(W[prefix + ".patch_embedding.weight"], emulate),
                 W.get(prefix + ".patch_embedding.bias"), stride=P)                     # [B,512,L/16,8]
    x = _r(x.reshape(B, x.shape[1], -1).transpose(1, 2), emulate, "embed")              # time-major tokens
    x = layernorm(x, W, prefix + ".layer_norm", cfg.layer_norm_eps, emulate)
    if (prefix + ".post_extract_proj.weight") in W:
        x = linear(x, W, prefix + ".post_extract_proj", emulate)
    n = x.shape[1]
    e = prefix + ".encoder"
    # pos_conv: weight-normed grouped Conv1d + SamePad + GELU (backbone.py:33-46,114-116)
    g_, v_ = W[e + ".pos_conv.0.weight_g"].float(), W[e + ".pos_conv.0.weight_v"].float()
    wn = g_ * v_ / v_.norm(p=2, dim=(0, 1), keepdim=True)                               # weight_norm dim=2
    pc = F.conv1d(_op(x, emulate).transpose(1, 2), _op(wn, emulate), W[e + ".pos_conv.0.bias"].float(), padding=cfg.conv_pos // 2,
                  groups=cfg.conv_pos_groups)
    if cfg.conv_pos % 2 == 0:
        pc = pc[:, :, :-1]
    pc = _gelu(pc).transpose(1, 2)
    x = layernorm(x + pc, W, e + ".layer_norm", cfg.layer_norm_eps, emulate)            # post-LN variant :118-119
    alpha = math.pow(2 * cfg.encoder_layers, 0.25) if cfg.deep_norm else 1.0
    buckets = rel_pos_bucket(n, n, cfg.num_buckets, cfg.max_distance)
    table = W[e + ".layers.0.self_attn.relative_attention_bias.weight"].float()          # shared (:78-81)
    pos_bias = table[buckets].permute(2, 0, 1)                                           # [H,n,n]
    scaling = d ** -0.5
    for i in range(cfg.encoder_layers):
        p = f"{e}.layers.{i}"
        q0 = linear(x, W, p + ".self_attn.q_proj", emulate)                             # un-scaled projection
        k = linear(x, W, p + ".self_attn.k_proj", emulate)
        v = linear(x, W, p + ".self_attn.v_proj", emulate)
        bias = pos_bias[None]
        if cfg.gru_rel_pos:                                                              # :650-662
            qh = q0.view(B, n, H, d).transpose(1, 2)                                     # [B,H,n,d]
            gl = F.linear(qh, W[p + ".self_attn.grep_linear.weight"].float(), W[p + ".self_attn.grep_linear.bias"].float())
            gsum = torch.sigmoid(gl.view(B, H, n, 2, 4).sum(-1))
            ga, gb = gsum[..., 0:1], gsum[..., 1:2]
            gate = ga * (gb * W[p + ".self_attn.grep_a"].float().view(1, H, 1, 1) - 1.0) + 2.0
            bias = gate * pos_bias[None]
        # (q*scaling/32 k^T - max)*32 + bias == q k^T scaling + bias up to a per-row shift (:513-515,623-667)
        o = _mha(q0, k, v, H, scaling, bias, emulate)
        a = linear(o, W, p + ".self_attn.out_proj", emulate, store=False)
        x = layernorm(x * alpha + a, W, p + ".self_attn_layer_norm", cfg.layer_norm_eps, emulate)
        f = _r(_gelu(linear(x, W, p + ".fc1", emulate, store=False)), emulate)
        f = linear(f, W, p + ".fc2", emulate, store=False)
        x = layernorm(x * alpha + f, W, p + ".final_layer_norm", cfg.layer_norm_eps, emulate)
    return x


def audio_encoder(audio: Tensor, W: WDict, cfg: BeatsConfig, emulate=None,
                  prefix: str = "model.audio_encoder.audio_encoder") -> Tensor:
    """AudioEncoder.forward 4-D branch (multimodal_encoder.py:174-186): [b,t,L,128] -> [b,t,n,768]."""
    b, t, L, m = audio.shape
    y = beats(audio.reshape(b * t, L, m), W, cfg, prefix, emulate)
    return y.reshape(b, t, y.shape[1], y.shape[2])


# =====================================================================================
# B.7 prepare_multimodal_inputs + generate      reference models/unified_arch.py:217-406
# =====================================================================================

SPECIAL_TOKENS = ['<image>', '<image_start>', '<image_end>', '<video>', '<video_start>', '<video_end>',
                  '<audio>', '<audio_start>', '<audio_end>', '<mask_start>', '<mask_end>']


def special_token_table(vocab_nums: int, mask_token_nums: int = 6) -> Dict[str, int]:
    """initialize_MM_tokenizer (unified_arch.py:409-459): ids vocab_nums.. in fixed order."""
    toks = SPECIAL_TOKENS + [f'<mask_{i}>' for i in range(mask_token_nums)]
    return {t: vocab_nums + i for i, t in enumerate(toks)}


@dataclass
class CrabConfig:
    decoder: DecoderConfig
    clip: Optional[ClipConfig] = None
    beats: Optional[BeatsConfig] = None
    qformer: QFormerConfig = field(default_factory=QFormerConfig)
    base_vocab: int = 32000            # len(tokenizer) before the 17 added tokens
    pad_token_id: int = 2              # tokenizer.pad_token = eos (quick_start.py:501-502)
    image_token_nums: int = 256


def encode_video(video: Tensor, W: WDict, cfg: CrabConfig, emulate=None, all_levels: bool = False):
    """encode_video live branch (unified_arch.py:144-149).  The reference runs the VLProjector on all
    three feature levels and consumes only [-1] (:290); all_levels=False skips the dead two."""
    feats = visual_encoder(video, W, cfg.clip, emulate)
    levels = feats if all_levels else feats[-1:]
    q = [vl_projector(f, W, cfg.qformer, cfg.image_token_nums, emulate=emulate) for f in levels]
    return feats, q


def encode_audio(audio: Tensor, W: WDict, cfg: CrabConfig, emulate=None) -> Tensor:
    return al_projector(audio_encoder(audio, W, cfg.beats, emulate), W, cfg.qformer, emulate=emulate)


def prepare_multimodal_inputs(batch_input_ids: Sequence[Tensor], batch_X_modals: Sequence[Dict[str, Tensor]],
                              W: WDict, cfg: CrabConfig, emulate=None) -> Dict[str, Tensor]:
    """unified_arch.py:217-406 (NTP branch: no multi-scale features): splice, left-pad, positions."""
    tab = special_token_table(cfg.base_vocab)
    keys = {tab['<image>']: '<image>', tab['<video>']: '<video>', tab['<audio>']: '<audio>'}
    emb = W["model.embed_tokens.weight"].float()
    seqs = []
    for ids, mod in zip(batch_input_ids, batch_X_modals):
        segs, pre = [], 0
        for pos in [i for i, t in enumerate(ids.tolist()) if t in keys]:
            segs.append(emb[ids[pre:pos]])
            key = keys[int(ids[pos])]
            if key == '<audio>':
                f = encode_audio(mod[key][None], W, cfg, emulate)[0]
            else:
                f = encode_video(mod[key][None], W, cfg, emulate)[1][-1][0]
            segs.append(f)
            pre = pos + 1
        segs.append(emb[ids[pre:]])
        seqs.append(torch.cat(segs, dim=0))
    L = max(s.shape[0] for s in seqs)
    pad_e = emb[cfg.pad_token_id]
    embeds, mask = [], []
    for s in seqs:
        n = L - s.shape[0]
        embeds.append(torch.cat([pad_e[None].expand(n, -1), s], dim=0))             # left pad :344-348
        mask.append(torch.cat([torch.zeros(n, dtype=torch.int32), torch.ones(s.shape[0], dtype=torch.int32)]))
    mask = torch.stack(mask)
    pos = torch.cumsum(mask, dim=-1) - 1
    pos[pos == -1] = 0                                                               # :372-373
    return {"inputs_embeds": _r(torch.stack(embeds), emulate), "attention_mask": mask, "position_ids": pos}


def generate(batch_input_ids, batch_X_modals, W: WDict, cfg: CrabConfig, max_new_tokens: int,
             eos_token_id: Optional[int] = None, min_new_tokens: int = 0, emulate=None):
    """UnifiedForCausalLM.generate (unified_llama.py:244-267) with greedy decoding forced.
    attention_mask / position_ids are NOT forwarded by the reference (:261-267), reproduced here."""
    inp = prepare_multimodal_inputs(batch_input_ids, batch_X_modals, W, cfg, emulate)
    return greedy_generate(inp["inputs_embeds"], W, cfg.decoder, max_new_tokens, eos_token_id,
                           cfg.pad_token_id, min_new_tokens, emulate)


# =====================================================================================
# B.8 SegModule (generate_avs pixel path)     reference models/multimodal_encoder.py:268-543, 891-1444
# =====================================================================================

def _ln2d(x: Tensor, W: WDict, prefix: str, eps: float = 1e-6) -> Tensor:
    """LayerNorm2d (:606-618): per-pixel LayerNorm over channels of [B,C,H,W]."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return W[prefix + ".weight"].float()[None, :, None, None] * x + W[prefix + ".bias"].float()[None, :, None, None]


def _dense_pe(G: Tensor, h: int, w: int) -> Tensor:
    """PositionEmbeddingRandom.forward (:825-839): [C,h,w], C = 2*G.shape[1]."""
    ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
    xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
    coords = torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)], dim=-1)
    c = (2 * coords - 1) @ G.float()
    c = 2 * math.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1).permute(2, 0, 1)


def _sam_attention(q: Tensor, k: Tensor, v: Tensor, W: WDict, p: str, heads: int = 8) -> Tensor:
    """Attention (:1333-1390): separate q/k/v Linear to internal_dim, heads, /sqrt(c_per_head), softmax, out Linear."""
    q = linear(q, W, p + ".q_proj")
    k = linear(k, W, p + ".k_proj")
    v = linear(v, W, p + ".v_proj")
    d = q.shape[-1] // heads
    o = _mha(q, k, v, heads, 1.0 / math.sqrt(d))
    return linear(o, W, p + ".out_proj")


def _torch_mha(q: Tensor, kv: Tensor, W: WDict, p: str, heads: int = 8) -> Tensor:
    """nn.MultiheadAttention(batch_first) with packed in_proj (query generator, :1397-1419)."""
    E = q.shape[-1]
    wi, bi = W[p + ".in_proj_weight"].float(), W[p + ".in_proj_bias"].float()
    qq = F.linear(q, wi[:E], bi[:E])
    kk = F.linear(kv, wi[E:2 * E], bi[E:2 * E])
    vv = F.linear(kv, wi[2 * E:], bi[2 * E:])
    o = _mha(qq, kk, vv, heads, 1.0 / math.sqrt(E // heads))
    return linear(o, W, p + ".out_proj")


def _query_generator(avs_query: Tensor, sparse: Tensor, W: WDict, p: str, num_layers: int) -> Tensor:
    """QueryGenerator.forward (:1441-1444): EVERY layer is fed the original avs_query, so only the last layer's
    output survives (reference quirk) -- reproduced."""
    query = None
    for l in range(num_layers):
        q = avs_query
        lp = f"{p}.layers.{l}"
        q = layernorm(q + _torch_mha(q, q, W, lp + ".self_attn"), W, lp + ".norm1", 1e-5)
        q = layernorm(q + _torch_mha(q, sparse, W, lp + ".cross_attn"), W, lp + ".norm2", 1e-5)
        f = linear(_gelu(linear(q, W, lp + ".ffn.0")), W, lp + ".ffn.2")
        query = layernorm(q + f, W, lp + ".norm3", 1e-5)
    return query


def _two_way_transformer(src: Tensor, pos: Tensor, tokens: Tensor, W: WDict, p: str, depth: int) -> Tuple[Tensor, Tensor]:
    """TwoWayTransformer.forward (:1209-1254) + TwoWayAttentionBlock.forward (:1299-1330)."""
    b, c, h, w = src.shape
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos.flatten(2).permute(0, 2, 1)
    queries, query_pe = tokens, tokens
    for i in range(depth):
        lp = f"{p}.layers.{i}"
        if i == 0:
            queries = _sam_attention(queries, queries, queries, W, lp + ".self_attn")
        else:
            q = queries + query_pe
            queries = queries + _sam_attention(q, q, queries, W, lp + ".self_attn")
        queries = layernorm(queries, W, lp + ".norm1", 1e-5)
        q, k = queries + query_pe, keys + key_pe
        queries = layernorm(queries + _sam_attention(q, k, keys, W, lp + ".cross_attn_token_to_image"), W, lp + ".norm2", 1e-5)
        m = linear(F.relu(linear(queries, W, lp + ".mlp.lin1")), W, lp + ".mlp.lin2")
        queries = layernorm(queries + m, W, lp + ".norm3", 1e-5)
        q, k = queries + query_pe, keys + key_pe
        keys = layernorm(keys + _sam_attention(k, q, queries, W, lp + ".cross_attn_image_to_token"), W, lp + ".norm4", 1e-5)
    q, k = queries + query_pe, keys + key_pe
    queries = layernorm(queries + _sam_attention(q, k, keys, W, p + ".final_attn_token_to_image"), W, p + ".norm_final_attn", 1e-5)
    return queries, keys


def _mlp3(x: Tensor, W: WDict, p: str) -> Tensor:
    x = F.relu(linear(x, W, p + ".layers.0"))
    x = F.relu(linear(x, W, p + ".layers.1"))
    return linear(x, W, p + ".layers.2")


def _predict_masks(img: Tensor, image_pe: Tensor, sparse: Tensor, dense: Tensor, level: int, prev: Optional[Tensor], task: str,
                   W: WDict, p: str, depth: int, qg_layers: int, nq: int) -> Tensor:
    """MaskDecoderMultiScale.predict_masks (:1083-1143)."""
    tokens = _query_generator(W[p + ".avs_query_tokens.weight"].float()[None], sparse, W, p + ".query_generator", qg_layers)
    tokens = tokens + W[p + ".level_embed.weight"].float()[level][None, None]
    src = img
    if level > 0:
        up = F.conv_transpose2d(src, W[p + ".upsample_2x.0.weight"].float(), W[p + ".upsample_2x.0.bias"].float(), stride=2)
        src = _gelu(_ln2d(up, W, p + ".upsample_2x.1"))
        pm = prev.mean(dim=1)
        src = (torch.sigmoid(pm)[:, None] + 1) * src
        image_pe = _dense_pe(W[p + ".pe1.positional_encoding_gaussian_matrix"], src.shape[2], src.shape[3])[None]
        dense = F.interpolate(dense.float(), size=src.shape[2:], mode="bilinear", align_corners=False)
    src = src + dense
    b, c, h, w = src.shape
    hs, keys = _two_way_transformer(src, image_pe, tokens, W, f"{p}.transformer.{level}", depth)
    t = _mlp3(hs[:, :nq], W, p + ".hyper_mlp")
    src = keys.transpose(1, 2).reshape(b, c, h, w)
    up = F.conv_transpose2d(src, W[p + ".output_upscaling.0.weight"].float(), W[p + ".output_upscaling.0.bias"].float(), stride=2)
    up = _gelu(_ln2d(up, W, p + ".output_upscaling.1"))
    b, c2, h2, w2 = up.shape
    masks = (t @ up.view(b, c2, h2 * w2)).view(b, -1, h2, w2)
    x = masks
    for i in range(3):
        x = F.conv2d(x, W[f"{p}.hyper_mlp_out.layers.{i}.weight"].float(), W[f"{p}.hyper_mlp_out.layers.{i}.bias"].float())
        if i < 2:
            x = F.relu(x)
    cls = p + (".avss_classifier.weight" if task == 'avss' else ".ms3_s4_classfier.weight")
    return F.conv2d(x, W[cls].float())


def seg_module(pred_embeddings: Tensor, feats: Sequence[Tensor], task_names: Sequence[str], W: WDict,
               prefix: str = "model.seg_module", emb_size: int = 16, low_res: int = 112, image_size: int = 224,
               scales: int = 2, toks: int = 3, depth: int = 2, qg_layers: int = 2, nq: int = 300) -> List[Tensor]:
    """SegModule.forward inference branch (:368-448): pred_embeddings [bs, scales*toks, D], feats = per level
    [bs, emb_size^2, C] -> list of [num_classes, image_size, image_size] (71 for 'avss', else 1)."""
    p = prefix
    e = linear(F.relu(linear(pred_embeddings.float(), W, p + ".text_hidden_fcs.0.0")), W, p + ".text_hidden_fcs.0.2")
    bs, n, dim = e.shape
    obj = n // (scales * toks)
    e = e.reshape(bs, obj, scales, toks, dim)
    fused = sum((1.0 / toks) * e[:, :, :, i] for i in range(toks))                       # multiseg_scalar: plain list, 1/3
    grid = []
    for f in feats:
        g = f.float().reshape(bs, -1, emb_size, emb_size, f.shape[-1]).permute(0, 1, 4, 2, 3)[:, 0]
        grid.append(g)
    grid = torch.stack(grid, dim=1)                                                        # [bs, level, C, s, s]
    pe = _dense_pe(W[p + ".pe_layer.positional_encoding_gaussian_matrix"], emb_size, emb_size)[None]
    out = []
    for i in range(bs):
        sparse = fused[i]                                                                  # [obj, scales, 256]
        dense = W[p + ".no_mask_embed.weight"].float().reshape(1, -1, 1, 1).expand(sparse.shape[0], -1, emb_size, emb_size)
        x = F.conv2d(grid[i], W[p + ".image_feature_neck.0.weight"].float())
        x = _ln2d(x, W, p + ".image_feature_neck.1")
        x = F.conv2d(x, W[p + ".image_feature_neck.2.weight"].float(), padding=1)
        img = _ln2d(x, W, p + ".image_feature_neck.3")                                    # [level, 256, s, s]
        low = None
        lm = None
        for l in range(scales):
            lm = _predict_masks(img[l][None], pe, sparse[:, l][:, None], dense, l, lm, task_names[i], W, p + ".mask_decoder",
                                depth, qg_layers, nq)
            up = (1.0 / scales) * F.interpolate(lm.float(), (low_res, low_res), mode="bilinear", align_corners=False)
            low = up if low is None else low + up
        out.append(F.interpolate(low, (image_size, image_size), mode="bilinear", align_corners=False)[0])
    return out
