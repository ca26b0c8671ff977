"""Encoder-side modules on the HIP kernels, mirroring reference `models/multimodal_encoder.py`:

  VisualEncoder  (multimodal_encoder.py:33-84)   CLIP ViT tower -> hidden states [14,22,23] minus CLS
  VLProjector    (:87-144)                       LayerNorm -> Q-Former (32 queries, cross-attn) -> MLP 768->D->D
  AudioEncoder   (:148-186)                      BEATs.extract_features(feature_only=True)
  ALProjector    (:189-262)                      LayerNorm -> Q-Former -> MLP
  build_mlp      (:25-30)

Class names, constructor argument meaning, forward signatures and parameter names (state-dict keys, as under
the reference's pinned transformers==4.37.2) are kept, so `finetune_weights.bin` / CLIP / BEATs checkpoints load
with load_state_dict.  The third-party towers the reference pulls from `transformers` (CLIPVisionModel) and from
`models/beats`, `models/Qformer.py` are re-expressed here as parameter containers + kernel launch sequences:
every GEMM, LayerNorm, attention, patch im2col etc. runs in libcrab_hip.so; PyTorch only owns the buffers.

Dead compute of the reference that is skipped (output-identical): CLIP layers above
max(select_layer_list) and `post_layernorm`; the text branch / LM head of the Q-Former.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from . import _lib, ops
from ._lib import GemmDesc
from .peft_hyper import PackedLinearGroup

BF16 = torch.bfloat16
NATIVE_ENC_LAYERS = True      # False: issue every launch of an encoder layer from Python (sequencer-equivalence tests, A/B runs)


def _native_layers() -> bool:
    """One C call per encoder layer - unless a per-launch profiler is attached (bench.py times individual GEMM launches through
    ops.gemm; the C sequencers issue the same launches but outside its view), like crab_amd/decoder.py does for the decoder layers."""
    return NATIVE_ENC_LAYERS and not ops.per_launch_profiling()


def _dense(lin) -> "_lib.Dense":
    """crab_dense over an nn.Linear-like holder (LinearP) or a bias-carrying PackedLinearGroup without adapter."""
    d = _lib.Dense()
    W = lin.W if isinstance(lin, PackedLinearGroup) else lin.weight
    d.W, d.ldw, d.N, d.K = W.data_ptr(), W.stride(0), W.shape[0], W.shape[1]
    d.bias = lin.bias.data_ptr() if lin.bias is not None else None
    return d


def _ln(ln) -> "_lib.LN":
    r = _lib.LN()
    r.w, r.b, r.eps = ln.weight.data_ptr(), ln.bias.data_ptr(), ln.eps
    r.fp32 = 1 if ln.weight.dtype == torch.float32 else 0
    assert ln.bias.dtype == ln.weight.dtype
    return r


class _EncScratch:
    """Caller-owned rows of crab_enc_io for M tokens of `width`, an FFN of `ffn`, qkv rows of `qkv_rows` x `qkv_cols`, V^T of `vt_elems`."""

    def __init__(self, M, width, ffn, qkv_rows, qkv_cols, vt_elems, device):
        e = lambda *s_, dt=BF16: torch.empty(s_, device=device, dtype=dt)
        # y: CLIP's mid-layer residual row x + attn / the pre-LN sums of the post-LN encoders - fp32 with ops.RESIDUAL_FP32 (crab_enc_io.x_fp32)
        self.a, self.y, self.att, self.f = e(M, width), e(M, width, dt=ops.RES_DTYPE), e(M, width), e(M, ffn)
        self.qkv = e(qkv_rows, qkv_cols)
        self.vt = torch.zeros((vt_elems,), device=device, dtype=BF16)

    def io(self, x, B, S) -> "_lib.EncIO":
        io = _lib.EncIO()
        io.x, io.a, io.y, io.qkv, io.att, io.f = (t.data_ptr() for t in (x, self.a, self.y, self.qkv, self.att, self.f))
        io.vt, io.vt_bytes = self.vt.data_ptr(), self.vt.numel() * 2
        ws = ops._splitk_workspace(x.device)
        io.workspace, io.workspace_bytes = ws.data_ptr(), ws.numel()
        io.B, io.S = B, S
        io.x_fp32 = 1 if self.y.dtype == torch.float32 else 0
        return io


def _p(t, device, *shape, fill=0.0, dtype=BF16):
    return nn.Parameter(torch.full(shape, fill, device=device, dtype=dtype), requires_grad=False)


class LinearP(nn.Module):
    """Parameter container for one nn.Linear (weight [out,in], optional bias); call = GEMM launch."""

    def __init__(self, in_f: int, out_f: int, device, bias: bool = True):
        super().__init__()
        self.in_features, self.out_features = in_f, out_f
        self.weight = _p(None, device, out_f, in_f)
        if bias:
            self.bias = _p(None, device, out_f)
        else:
            self.register_parameter("bias", None)

    def forward(self, x, act="none", residual=None, res_scale=1.0, out=None, out_fp32=False):
        return ops.gemm(x, self.weight, bias=self.bias, act=act, residual=residual, res_scale=res_scale, out=out, out_fp32=out_fp32)


class LayerNormP(nn.Module):
    """torch.nn.LayerNorm's parameters; fp32 by default (ops.NORM_DTYPE: not matrix operands, see crab_amd/ops.py), so a checkpoint's fp32
    weight / bias load unrounded."""

    def __init__(self, dim: int, eps: float, device):
        super().__init__()
        self.weight = _p(None, device, dim, fill=1.0, dtype=ops.NORM_DTYPE)
        self.bias = _p(None, device, dim, dtype=ops.NORM_DTYPE)
        self.eps = eps

    def forward(self, x, out=None):
        return ops.layernorm(x, self.weight, self.bias, self.eps, out=out)


def _attention(q, k, vt_src, out, *, B, H, Sq, Skv, d, ldq, ldk, q_off=0, k_off=0, scale, bias=None, gate=None,
               split_src=None, split_H=0, split_ld=None):
    """Bidirectional MHA over token-major projections.  q/k are read in place with strides; V^T is materialised
    from the packed projection `split_src` ([B*Skv, split_ld], v heads after split_H q-heads and H k-heads)."""
    dev = q.device
    Sp = (Skv + 7) // 8 * 8
    vt = torch.zeros((B, H, d, Sp), device=dev, dtype=BF16) if Sp != Skv else torch.empty((B, H, d, Sp), device=dev, dtype=BF16)
    ops.qkv_rope_split(split_src, None, None, None, vt, B, Skv, split_H, H, d, 1, 0, None)
    qv = q[:, q_off:] if q_off else q
    kv = k[:, k_off:] if k_off else k
    ops.attn_fwd(qv, kv, vt, out, q_strides=(Sq * ldq, d, ldq), k_strides=(Skv * ldk, d, ldk),
                 vt_strides=(H * d * Sp, d * Sp, Sp), o_strides=(Sq * H * d, H * d), B=B, H=H, Hk=H, Sq=Sq, Skv=Skv,
                 head_dim=d, scale=scale, bias=bias, gate=gate)
    return out


# =====================================================================================================
# CLIP ViT tower  (HF CLIPVisionModel)
# =====================================================================================================

class _ClipEmbeddings(nn.Module):
    def __init__(self, D, patch, n_pos, device):
        super().__init__()
        self.class_embedding = _p(None, device, D)
        self.patch_embedding = nn.Module()
        self.patch_embedding.weight = _p(None, device, D, 3, patch, patch)        # Conv2d(3,D,k=s=patch,bias=False)
        self.position_embedding = nn.Module()
        self.position_embedding.weight = _p(None, device, n_pos, D)


class _ClipAttention(nn.Module):
    def __init__(self, D, device):
        super().__init__()
        self._qkv = PackedLinearGroup(["q_proj", "k_proj", "v_proj"], D, [D, D, D], True, device)
        self.q_proj, self.k_proj, self.v_proj = self._qkv.linears
        self.out_proj = LinearP(D, D, device)


class _ClipMLP(nn.Module):
    def __init__(self, D, I, device):
        super().__init__()
        self.fc1 = LinearP(D, I, device)
        self.fc2 = LinearP(I, D, device)


class _ClipLayer(nn.Module):
    def __init__(self, D, I, eps, device):
        super().__init__()
        self.self_attn = _ClipAttention(D, device)
        self.layer_norm1 = LayerNormP(D, eps, device)
        self.mlp = _ClipMLP(D, I, device)
        self.layer_norm2 = LayerNormP(D, eps, device)


class _ClipEncoder(nn.Module):
    def __init__(self, n, D, I, eps, device):
        super().__init__()
        self.layers = nn.ModuleList([_ClipLayer(D, I, eps, device) for _ in range(n)])


class _ClipVisionTransformer(nn.Module):
    def __init__(self, cfg: Dict, device):
        super().__init__()
        D = cfg["hidden_size"]
        n_pos = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
        self.embeddings = _ClipEmbeddings(D, cfg["patch_size"], n_pos, device)
        self.pre_layrnorm = LayerNormP(D, cfg["layer_norm_eps"], device)          # (sic) HF attribute name
        self.encoder = _ClipEncoder(cfg["num_hidden_layers"], D, cfg["intermediate_size"], cfg["layer_norm_eps"], device)
        self.post_layernorm = LayerNormP(D, cfg["layer_norm_eps"], device)       # present in checkpoints, dead on this path


CLIP_VIT_L14 = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                    image_size=224, patch_size=14, layer_norm_eps=1e-5)


class CLIPVisionModel(nn.Module):
    """Parameter-compatible stand-in for transformers.CLIPVisionModel (keys `vision_model.*`)."""

    def __init__(self, config: Optional[Dict] = None, device="cuda"):
        super().__init__()
        self.config = dict(CLIP_VIT_L14 if config is None else config)
        self.vision_model = _ClipVisionTransformer(self.config, device)
        self._pw = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """Accept both serialisations of the HF tower: `vision_model.<...>` (transformers 4.37.2, the reference's pin, and the checkpoint
        files on the hub) and the flattened `<...>` a transformers 5.x CLIPVisionModel.state_dict() gives (tests/golden/ckpt_manifest.npz);
        `embeddings.position_ids` (a persistent buffer in checkpoints written before transformers 4.31) is not a parameter here."""
        for k in [k for k in state_dict if k.startswith(prefix) and not k.startswith(prefix + "vision_model.")]:
            state_dict[prefix + "vision_model." + k[len(prefix):]] = state_dict.pop(k)
        state_dict.pop(prefix + "vision_model.embeddings.position_ids", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _patch_weight(self):
        w = self.vision_model.embeddings.patch_embedding.weight
        key = (w.data_ptr(), w._version)
        if self._pw is None or self._pw[0] != key:
            D = w.shape[0]
            K = w[0].numel()
            Kp = (K + 31) // 32 * 32
            pw = torch.zeros((D, Kp), device=w.device, dtype=BF16)
            ops.copy_rows(w.reshape(D, K), pw, D, K)                       # zero-padded K for 16-byte rows
            self._pw = (key, pw)
        return self._pw[1]

    def hidden_states(self, pixel_values: torch.Tensor, upto: int, keep: Optional[Sequence[int]] = None) -> Dict[int, torch.Tensor]:
        """pixel_values [N,3,H,W] (fp32 or bf16) -> {l: h_l [N, 1+P, D] bf16} for l in keep (default: all 0..upto)."""
        c = self.config
        vm = self.vision_model
        N = pixel_values.shape[0]
        D, Hh, ps = c["hidden_size"], c["num_attention_heads"], c["patch_size"]
        d = D // Hh
        P = (c["image_size"] // ps) ** 2
        T = P + 1
        pw = self._patch_weight()
        patches = ops.im2col_patch(pixel_values, ps, pw.shape[1])                       # [N*P, Kp]
        pe = ops.gemm(patches, pw)                                                      # conv14/14, no bias
        h = ops.clip_embed_ln(pe, vm.embeddings.class_embedding, vm.embeddings.position_embedding.weight,
                              vm.pre_layrnorm.weight, vm.pre_layrnorm.bias, N, P, D, vm.pre_layrnorm.eps)
        keep = set(range(upto + 1)) if keep is None else set(keep)
        hs = {0: h.view(N, T, D)} if 0 in keep else {}
        M = N * T
        f32 = ops.RESIDUAL_FP32
        if f32 and upto > 0:
            # the residual stream of the pre-LN tower is fp32 from here on; the kept hidden states go back to bf16 (what the projectors read)
            h16, h = h, torch.empty((M, D), device=h.device, dtype=torch.float32)
            ops.cast_rows(h16, h, M, D)
            if 0 in hs:
                hs[0] = h16.view(N, T, D)

        def keep_state(x):
            if not f32:
                return x
            o = torch.empty((M, D), device=x.device, dtype=BF16)
            ops.cast_rows(x, o, M, D)
            return o

        if _native_layers():
            # one C call per layer (crab_clip_layer, csrc/encoder_layers.hip): the launches below, in the same order, x updated in place
            sc = _EncScratch(M, D, c["intermediate_size"], M, 3 * D, N * D * ((T + 7) // 8 * 8), h.device)
            if 0 in hs and upto > 0 and not f32:
                hs[0] = h.clone().view(N, T, D)                # h is updated in place from here on
            io = sc.io(h, N, T)
            for i in range(upto):
                L = vm.encoder.layers[i]
                w = _lib.ClipLayerW()
                w.ln1, w.ln2, w.qkv, w.out = _ln(L.layer_norm1), _ln(L.layer_norm2), _dense(L.self_attn._qkv), _dense(L.self_attn.out_proj)
                w.fc1, w.fc2, w.H = _dense(L.mlp.fc1), _dense(L.mlp.fc2), Hh
                ops.enc_layer("clip", w, io, h.device)
                if i + 1 in keep:
                    hs[i + 1] = keep_state(h).view(N, T, D) if f32 else (h.clone().view(N, T, D) if i + 1 < upto else h.view(N, T, D))
            return hs
        a = torch.empty((M, D), device=h.device, dtype=BF16)
        qkv = torch.empty((M, 3 * D), device=h.device, dtype=BF16)
        att = torch.empty((M, D), device=h.device, dtype=BF16)
        f1 = torch.empty((M, c["intermediate_size"]), device=h.device, dtype=BF16)
        for i in range(upto):
            L = vm.encoder.layers[i]
            L.layer_norm1(h, out=a)
            L.self_attn._qkv(a, out=qkv)
            _attention(qkv, qkv, None, att, B=N, H=Hh, Sq=T, Skv=T, d=d, ldq=3 * D, ldk=3 * D, k_off=D, scale=d ** -0.5,
                       split_src=qkv, split_H=Hh)
            hn = L.self_attn.out_proj(att, residual=h, out_fp32=f32)
            L.layer_norm2(hn, out=a)
            L.mlp.fc1(a, act="quick_gelu", out=f1)
            h = L.mlp.fc2(f1, residual=hn, out_fp32=f32)
            if i + 1 in keep:
                hs[i + 1] = keep_state(h).view(N, T, D)
        return hs


def build_mlp(depth, hidden_size, output_hidden_size, device="cuda"):
    """multimodal_encoder.py:25-30: Linear, then (GELU, Linear) x (depth-1); keys `0.weight`, `2.weight`, ..."""
    mods = [LinearP(hidden_size, output_hidden_size, device)]
    for _ in range(1, depth):
        mods.append(nn.Identity())           # slot of nn.GELU(): fused into the previous GEMM's epilogue
        mods.append(LinearP(output_hidden_size, output_hidden_size, device))
    return nn.Sequential(*mods)


def _run_mlp(mlp: nn.Sequential, x):
    lin = [m for m in mlp if isinstance(m, LinearP)]
    for i, m in enumerate(lin):
        x = m(x, act="gelu" if i + 1 < len(lin) else "none")
    return x


class VisualEncoder(nn.Module):
    """multimodal_encoder.py:33-84."""

    def __init__(self, model_name_or_path=None, select_layer_list=[-11, -2, -1], select_feature='patch', config=None,
                 device="cuda"):
        super().__init__()
        self.select_layer_list = list(select_layer_list)
        self.select_feature = select_feature
        if select_feature not in ('patch', 'cls_patch'):
            raise ValueError(f'Unexpected select feature: {select_feature}')
        # model_name_or_path is accepted for signature compatibility; weights arrive through load_state_dict
        self.vision_tower = CLIPVisionModel(config, device=device)
        # multimodal_encoder.py:46 exposes the tower's CLIPImageProcessor to the dataset code (quick_start.py:561); here it is
        # the device mirror (crab_amd/frontend.py), bit-exact with the Pillow path for uint8 RGB input
        from .frontend import CLIPImageProcessor
        size = int(self.vision_tower.config.get("image_size", 224))
        self.image_processor = CLIPImageProcessor(size=size, crop_size=size, device=device)

    def _layers(self):
        n = self.vision_tower.config["num_hidden_layers"] + 1
        return [l if l >= 0 else n + l for l in self.select_layer_list]

    @torch.no_grad()
    def encode_video(self, video) -> List[torch.Tensor]:
        b, t, c, h, w = video.shape
        sel = self._layers()
        hs = self.vision_tower.hidden_states(video.reshape(b * t, c, h, w), max(sel), keep=sel)
        feats = []
        for l in sel:
            x = hs[l]                                                          # [bt, 1+P, D]
            if self.select_feature == 'patch':
                N, T, D = x.shape
                f = torch.empty((N, T - 1, D), device=x.device, dtype=BF16)
                ops.copy_rows_batched(x[:, 1:], D, T * D, f, D, (T - 1) * D, N, T - 1, D)      # drop CLS
                x = f
            feats.append(x)
        return feats

    def forward(self, video) -> List[torch.Tensor]:
        b, t = video.shape[:2]
        return [f.reshape(b, t * f.shape[1], f.shape[2]) for f in self.encode_video(video)]


# =====================================================================================================
# Q-Former (models/Qformer.py; BLIP-2).  Only `.bert` with query_embeds is on the path.
# =====================================================================================================

BERT_BASE = dict(hidden_size=768, num_attention_heads=12, intermediate_size=3072, layer_norm_eps=1e-12,
                 vocab_size=30522, max_position_embeddings=512)


class _BertSelfAttention(nn.Module):
    def __init__(self, h, kv_width, device, cross: bool):
        super().__init__()
        self.query = LinearP(h, h, device)
        # key/value packed (one GEMM over the encoder states); q kept separate: different input on the cross path
        self._kv = PackedLinearGroup(["key", "value"], kv_width, [h, h], True, device)
        self.key, self.value = self._kv.linears


class _BertSelfOutput(nn.Module):
    def __init__(self, h, eps, device, in_f=None):
        super().__init__()
        self.dense = LinearP(in_f or h, h, device)
        self.LayerNorm = LayerNormP(h, eps, device)


class _BertAttention(nn.Module):
    def __init__(self, h, kv_width, eps, device, cross):
        super().__init__()
        setattr(self, "self", _BertSelfAttention(h, kv_width, device, cross))
        self.output = _BertSelfOutput(h, eps, device)


class _BertIntermediate(nn.Module):
    def __init__(self, h, i, device):
        super().__init__()
        self.dense = LinearP(h, i, device)


class _BertLayer(nn.Module):
    def __init__(self, cfg, enc_width, device):
        super().__init__()
        h, i, eps = cfg["hidden_size"], cfg["intermediate_size"], cfg["layer_norm_eps"]
        self.attention = _BertAttention(h, h, eps, device, False)
        self.crossattention = _BertAttention(h, enc_width, eps, device, True)        # cross_attention_freq = 1
        self.intermediate = _BertIntermediate(h, i, device)                          # text FFN: in checkpoints, unused
        self.output = _BertSelfOutput(h, eps, device, in_f=i)
        self.intermediate_query = _BertIntermediate(h, i, device)
        self.output_query = _BertSelfOutput(h, eps, device, in_f=i)


class _BertEmbeddings(nn.Module):
    def __init__(self, cfg, device):
        super().__init__()
        h = cfg["hidden_size"]
        self.word_embeddings = nn.Module()
        self.word_embeddings.weight = _p(None, device, cfg["vocab_size"], h)          # unused (query_embeds only)
        self.position_embeddings = nn.Module()
        self.position_embeddings.weight = _p(None, device, cfg["max_position_embeddings"], h)
        self.LayerNorm = LayerNormP(h, cfg["layer_norm_eps"], device)


class _BertEncoder(nn.Module):
    def __init__(self, cfg, n_layers, enc_width, device):
        super().__init__()
        self.layer = nn.ModuleList([_BertLayer(cfg, enc_width, device) for _ in range(n_layers)])


class BertModel(nn.Module):
    def __init__(self, cfg, n_layers, enc_width, device):
        super().__init__()
        self.config = cfg
        self.embeddings = _BertEmbeddings(cfg, device)
        self.encoder = _BertEncoder(cfg, n_layers, enc_width, device)

    def run(self, query: torch.Tensor, enc: torch.Tensor, B: int, nq: int, m: int) -> torch.Tensor:
        """query [nq,h] (shared learned tokens) or [B*nq,h]; enc [B*m, enc_width] (already layer-normed) -> [B*nq,h]."""
        c = self.config
        h, H = c["hidden_size"], c["num_attention_heads"]
        d = h // H
        dev = enc.device
        z0 = self.embeddings.LayerNorm(query)                                        # Qformer.py:105-108
        if z0.shape[0] == nq and B > 1:
            z = torch.empty((B * nq, h), device=dev, dtype=BF16)
            ops.copy_rows_batched(z0, h, 0, z, h, nq * h, B, nq, h)                  # broadcast the query tokens
        else:
            z = z0
        if _native_layers():
            # one C call per layer (crab_qformer_layer): self-attention, cross-attention to `enc`, query FFN; z updated in place
            i_ = self.encoder.layer[0].intermediate_query.dense.weight.shape[0]
            rows = max(B * nq, B * m)
            sc = _EncScratch(B * nq, h, i_, rows, 2 * h, B * h * ((max(nq, m) + 7) // 8 * 8), dev)
            io = sc.io(z, B, nq)
            io.enc, io.enc_rows = enc.data_ptr(), m
            for L in self.encoder.layer:
                sa, ca = getattr(L.attention, "self"), getattr(L.crossattention, "self")
                w = _lib.QformerLayerW()
                w.sq, w.skv, w.so, w.sln = _dense(sa.query), _dense(sa._kv), _dense(L.attention.output.dense), _ln(L.attention.output.LayerNorm)
                w.cq, w.ckv, w.co, w.cln = _dense(ca.query), _dense(ca._kv), _dense(L.crossattention.output.dense), _ln(L.crossattention.output.LayerNorm)
                w.iq, w.oq, w.oln, w.H = _dense(L.intermediate_query.dense), _dense(L.output_query.dense), _ln(L.output_query.LayerNorm), H
                ops.enc_layer("qformer", w, io, dev)
            return z
        att = torch.empty((B * nq, h), device=dev, dtype=BF16)
        scale = 1.0 / math.sqrt(d)
        for L in self.encoder.layer:
            sa = getattr(L.attention, "self")
            q = sa.query(z)
            kv = sa._kv(z)                                                           # [B*nq, 2h]
            _attention(q, kv, None, att, B=B, H=H, Sq=nq, Skv=nq, d=d, ldq=h, ldk=2 * h, scale=scale, split_src=kv, split_H=0)
            f32 = ops.RESIDUAL_FP32                                                  # fp32 pre-LN sums (crab_enc_io.x_fp32)
            z = L.attention.output.LayerNorm(L.attention.output.dense(att, residual=z, out_fp32=f32))          # :287-291
            ca = getattr(L.crossattention, "self")
            q = ca.query(z)
            kv = ca._kv(enc)                                                         # [B*m, 2h]
            _attention(q, kv, None, att, B=B, H=H, Sq=nq, Skv=m, d=d, ldq=h, ldk=2 * h, scale=scale, split_src=kv, split_H=0)
            z = L.crossattention.output.LayerNorm(L.crossattention.output.dense(att, residual=z, out_fp32=f32))
            f = L.intermediate_query.dense(z, act="gelu")                            # :483-486
            z = L.output_query.LayerNorm(L.output_query.dense(f, residual=z, out_fp32=f32))
        return z


class _BertLMHeadModel(nn.Module):
    """Holds `.bert` (and the unused `.cls` head keys are simply absent: load with strict=False like the reference)."""

    def __init__(self, cfg, n_layers, enc_width, device):
        super().__init__()
        self.bert = BertModel(cfg, n_layers, enc_width, device)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """The reference's BertLMHeadModel also owns `cls.predictions.*` (the 30522-way LM head: trainable, so finetune_weights.bin carries it,
        utils/deepspeed_utils.py:56-59) and the `bert.embeddings.position_ids` buffer; neither is ever read on the path (only `.bert` with
        query_embeds runs, models/multimodal_encoder.py:119-144).  They are consumed here instead of being reported as unexpected, so that a
        reference checkpoint loads with strict=True; they are not kept (24 M dead parameters per projector)."""
        for k in [k for k in state_dict if k.startswith(prefix + "cls.")]:
            state_dict.pop(k)
        state_dict.pop(prefix + "bert.embeddings.position_ids", None)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class VLProjector(nn.Module):
    """multimodal_encoder.py:87-144."""

    def __init__(self, bert_ckpt_path=None, hidden_size=1024, image_token_nums=256, num_query_token=32, num_hidden_layers=2,
                 d_model=3584, depth=2, bert_config=None, device="cuda"):
        super().__init__()
        cfg = dict(BERT_BASE if bert_config is None else bert_config)
        self.num_query_token = num_query_token
        self.image_token_nums = image_token_nums
        self.visual_ln = LayerNormP(hidden_size, 1e-5, device)
        self.visual_Qformer = _BertLMHeadModel(cfg, num_hidden_layers, hidden_size, device)
        self.visual_query_tokens = _p(None, device, 1, num_query_token, cfg["hidden_size"])
        self.visual_proj = build_mlp(depth, cfg["hidden_size"], d_model, device)

    def forward(self, visual_feature):
        b, tn, dim = visual_feature.shape
        n = self.image_token_nums
        t = tn // n
        x = self.visual_ln(visual_feature.reshape(b * tn, dim))
        z = self.visual_Qformer.bert.run(self.visual_query_tokens[0], x, b * t, self.num_query_token, n)
        y = _run_mlp(self.visual_proj, z)
        return y.view(b, t * self.num_query_token, -1)


class ALProjector(nn.Module):
    """multimodal_encoder.py:189-262 (both the 4-D [b,t,n,d] and 3-D [b,n,d] branches)."""

    def __init__(self, bert_ckpt_path=None, hidden_size=768, num_query_token=32, num_hidden_layers=2, d_model=3584, depth=2,
                 bert_config=None, device="cuda"):
        super().__init__()
        cfg = dict(BERT_BASE if bert_config is None else bert_config)
        self.audio_ln = LayerNormP(hidden_size, 1e-5, device)
        self.num_query_token = num_query_token
        self.audio_Qformer = _BertLMHeadModel(cfg, num_hidden_layers, hidden_size, device)
        self.audio_query_tokens = _p(None, device, 1, num_query_token, cfg["hidden_size"])
        self.audio_proj = build_mlp(depth, cfg["hidden_size"], d_model, device)

    def forward(self, audio_feature):
        if audio_feature.dim() == 4:
            b, t, n, d = audio_feature.shape
        else:
            b, n, d = audio_feature.shape
            t = 1
        x = self.audio_ln(audio_feature.reshape(b * t * n, d))
        z = self.audio_Qformer.bert.run(self.audio_query_tokens[0], x, b * t, self.num_query_token, n)
        y = _run_mlp(self.audio_proj, z)
        return y.view(b, t * self.num_query_token, -1)


# =====================================================================================================
# BEATs  (models/beats/BEATs.py:134-182, backbone.py)
# =====================================================================================================

class BEATsConfig:
    """models/beats/BEATs.py:26-69: defaults + dict update (the real values come from ckpt['cfg'])."""

    def __init__(self, cfg=None):
        self.input_patch_size = 16
        self.embed_dim = 512
        self.conv_bias = False
        self.encoder_layers = 12
        self.encoder_embed_dim = 768
        self.encoder_ffn_embed_dim = 3072
        self.encoder_attention_heads = 12
        self.activation_fn = "gelu"
        self.layer_norm_first = False
        self.deep_norm = True
        self.conv_pos = 128
        self.conv_pos_groups = 16
        self.relative_position_embedding = True
        self.num_buckets = 320
        self.max_distance = 800
        self.gru_rel_pos = True
        if cfg is not None:
            self.__dict__.update(cfg)


class _BeatsMHA(nn.Module):
    def __init__(self, E, H, num_buckets, device, own_table: bool):
        super().__init__()
        self._qkv = PackedLinearGroup(["q_proj", "k_proj", "v_proj"], E, [E, E, E], True, device)
        self.q_proj, self.k_proj, self.v_proj = self._qkv.linears
        self.out_proj = LinearP(E, E, device)
        self.grep_linear = LinearP(E // H, 8, device)
        self.grep_a = _p(None, device, 1, H, 1, 1, fill=1.0)
        if own_table:
            self.relative_attention_bias = nn.Module()
            self.relative_attention_bias.weight = _p(None, device, num_buckets, H)


class _BeatsLayer(nn.Module):
    def __init__(self, cfg: BEATsConfig, device, first: bool):
        super().__init__()
        E, F_, H = cfg.encoder_embed_dim, cfg.encoder_ffn_embed_dim, cfg.encoder_attention_heads
        self.self_attn = _BeatsMHA(E, H, cfg.num_buckets, device, first)
        self.self_attn_layer_norm = LayerNormP(E, 1e-5, device)
        self.fc1 = LinearP(E, F_, device)
        self.fc2 = LinearP(F_, E, device)
        self.final_layer_norm = LayerNormP(E, 1e-5, device)


class _BeatsPosConv(nn.Module):
    """nn.utils.weight_norm(Conv1d(E,E,k,groups), dim=2): keys `bias`, `weight_g` [1,1,k], `weight_v` [E,E/G,k]."""

    def __init__(self, E, G, k, device):
        super().__init__()
        self.bias = _p(None, device, E)
        self.weight_g = _p(None, device, 1, 1, k, fill=1.0)
        self.weight_v = _p(None, device, E, E // G, k, fill=1.0)


class _BeatsEncoder(nn.Module):
    def __init__(self, cfg: BEATsConfig, device):
        super().__init__()
        E = cfg.encoder_embed_dim
        if E % cfg.conv_pos_groups or (E // cfg.conv_pos_groups) % 8 or E % cfg.encoder_attention_heads or (E // cfg.encoder_attention_heads) not in (32, 64, 128):
            # the grouped positional convolution runs as one batched GEMM per (group, sequence) whose rows are the group's channels: 16-byte
            # LDS-DMA pieces need 8 of them (BEATs iter3+: 768 / 16 = 48); the attention kernels cover head sizes 32 / 64 / 128 (iter3+: 64)
            raise NotImplementedError(f"BEATs on the HIP path: encoder_embed_dim / conv_pos_groups must be a multiple of 8 and the head size one of "
                                      f"32 / 64 / 128 (got {E} / {cfg.conv_pos_groups}, {cfg.encoder_attention_heads} heads)")
        self.pos_conv = nn.Sequential(_BeatsPosConv(E, cfg.conv_pos_groups, cfg.conv_pos, device))
        self.layers = nn.ModuleList([_BeatsLayer(cfg, device, i == 0) for i in range(cfg.encoder_layers)])
        self.layer_norm = LayerNormP(E, 1e-5, device)


class BEATs(nn.Module):
    def __init__(self, cfg: BEATsConfig, device="cuda"):
        super().__init__()
        if not (cfg.deep_norm or not cfg.layer_norm_first) or cfg.layer_norm_first:
            raise NotImplementedError("only the post-LN (deep_norm) BEATs variant Crab ships is implemented")
        if cfg.activation_fn != "gelu" or not cfg.relative_position_embedding:
            raise NotImplementedError("BEATs variant outside the Crab checkpoint family")
        self.cfg = cfg
        P = cfg.input_patch_size
        self.patch_embedding = nn.Module()
        self.patch_embedding.weight = _p(None, device, cfg.embed_dim, 1, P, P)
        if cfg.conv_bias:
            self.patch_embedding.bias = _p(None, device, cfg.embed_dim)
        self.layer_norm = LayerNormP(cfg.embed_dim, 1e-5, device)
        self.post_extract_proj = LinearP(cfg.embed_dim, cfg.encoder_embed_dim, device) \
            if cfg.embed_dim != cfg.encoder_embed_dim else None
        self.encoder = _BeatsEncoder(cfg, device)
        self._pc = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # layers 1.. alias layer 0's relative_attention_bias (backbone.py:78-81): drop the duplicate keys
        for k in [k for k in state_dict if k.startswith(prefix + "encoder.layers.") and k.endswith("relative_attention_bias.weight")
                  and not k.startswith(prefix + "encoder.layers.0.")]:
            state_dict.pop(k)
        # the AudioSet classifier head of the fine-tuned checkpoint (BEATs.py: `predictor`, cfg.finetuned_model): never evaluated by
        # extract_features(feature_only=True), the only call on the path (models/multimodal_encoder.py:167-171)
        for k in [k for k in state_dict if k.startswith(prefix + "predictor.")]:
            state_dict.pop(k)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _posconv_weight(self):
        """weight_norm(dim=2) folded once: w[co,ci,k] = g[k] * v[co,ci,k] / ||v[:,:,k]||, re-laid out as
        [G][co_in_group][(k, ci)] for the sliding-window GEMM (one-off weight preparation)."""
        pc = self.encoder.pos_conv[0]
        key = (pc.weight_v.data_ptr(), pc.weight_v._version, pc.weight_g._version)
        if self._pc is None or self._pc[0] != key:
            v = pc.weight_v.float()
            w = pc.weight_g.float() * v / v.norm(p=2, dim=(0, 1), keepdim=True)
            E, cg, k = w.shape
            G = E // cg
            w = w.view(G, cg, cg, k).permute(0, 1, 3, 2).reshape(G, cg, k * cg).contiguous().to(BF16)
            self._pc = (key, w)
        return self._pc[1]

    @torch.no_grad()
    def extract_features(self, source, padding_mask=None, feature_only=True, **_):
        """BEATs.extract_features (BEATs.py:134-182) with the all-False padding mask Crab passes: [B,L,mel] -> [B,n,E]."""
        c = self.cfg
        B, L, mel = source.shape
        P, E, H = c.input_patch_size, c.encoder_embed_dim, c.encoder_attention_heads
        d = E // H
        dev = source.device
        pw = self.patch_embedding.weight.reshape(c.embed_dim, P * P)
        patches = ops.im2col_patch(source.reshape(B, 1, L, mel), P, P * P)
        n = (L // P) * (mel // P)
        f = ops.gemm(patches, pw, bias=getattr(self.patch_embedding, "bias", None))
        x = self.layer_norm(f)
        if self.post_extract_proj is not None:
            x = self.post_extract_proj(x)                                               # [B*n, E]
        enc = self.encoder
        G, Kc = c.conv_pos_groups, c.conv_pos
        cg = E // G
        xp = ops.beats_posconv_pad(x, B, n, E, G, Kc)                                   # [G][B][n+Kc-1][cg]
        w = self._posconv_weight()
        y = torch.empty((B * n, E), device=dev, dtype=ops.RES_DTYPE)                    # the pre-LN sum x + gelu(pos_conv(x)): fp32 (ops.RESIDUAL_FP32)
        g = GemmDesc()
        npad = n + Kc - 1
        g.A, g.B, g.C, g.bias, g.R = xp.data_ptr(), w.data_ptr(), y.data_ptr(), enc.pos_conv[0].bias.data_ptr(), x.data_ptr()
        g.lda, g.ldb, g.ldc, g.ldr = cg, Kc * cg, E, E
        g.M, g.N, g.K = n, cg, Kc * cg
        g.act, g.c_fp32, g.res_scale = 1, 1 if y.dtype == torch.float32 else 0, 1.0     # x + gelu(conv + bias)
        g.batch, g.nb0 = G * B, B
        g.sA0, g.sA1 = npad * cg, B * npad * cg
        g.sB0, g.sB1 = 0, cg * Kc * cg
        g.sC0, g.sC1 = n * E, cg
        g.sR0, g.sR1 = n * E, cg
        g.sBias0, g.sBias1 = 0, cg
        ops.gemm_desc(g, dev.index or 0)
        x = enc.layer_norm(y)
        alpha = math.pow(2 * c.encoder_layers, 0.25) if c.deep_norm else 1.0
        table = enc.layers[0].self_attn.relative_attention_bias.weight
        bias = ops.beats_relpos_bias(table, n, H, c.num_buckets, c.max_distance)        # once per forward (:131-137)
        if _native_layers():
            # one C call per layer (crab_beats_layer): gated relative-position attention + FFN, post-LN deep-norm; x updated in place
            sc = _EncScratch(B * n, E, c.encoder_ffn_embed_dim, B * n, 3 * E, B * E * ((n + 7) // 8 * 8), dev)
            io = sc.io(x, B, n)
            gate = torch.empty((B, H, n), device=dev, dtype=torch.float32)
            io.bias, io.gate = bias.data_ptr(), gate.data_ptr()
            for L_ in enc.layers:
                a = L_.self_attn
                w = _lib.BeatsLayerW()
                w.qkv, w.out, w.fc1, w.fc2 = _dense(a._qkv), _dense(a.out_proj), _dense(L_.fc1), _dense(L_.fc2)
                w.ln_attn, w.ln_final, w.H, w.alpha = _ln(L_.self_attn_layer_norm), _ln(L_.final_layer_norm), H, alpha
                if c.gru_rel_pos:
                    ga = a.grep_a.reshape(-1)
                    w.grep_w, w.grep_b, w.grep_a = a.grep_linear.weight.data_ptr(), a.grep_linear.bias.data_ptr(), ga.data_ptr()
                ops.enc_layer("beats", w, io, dev)
            return x.view(B, n, E), padding_mask
        att = torch.empty((B * n, E), device=dev, dtype=BF16)
        for L_ in enc.layers:
            a = L_.self_attn
            qkv = a._qkv(x)
            gate = ops.beats_gru_gate(qkv, a.grep_linear.weight, a.grep_linear.bias, a.grep_a.reshape(-1), B, n, H, d) \
                if c.gru_rel_pos else None
            _attention(qkv, qkv, None, att, B=B, H=H, Sq=n, Skv=n, d=d, ldq=3 * E, ldk=3 * E, k_off=E, scale=d ** -0.5,
                       bias=bias, gate=gate, split_src=qkv, split_H=H)
            f32 = ops.RESIDUAL_FP32                                                     # fp32 pre-LN sums (crab_enc_io.x_fp32)
            x = L_.self_attn_layer_norm(a.out_proj(att, residual=x, res_scale=alpha, out_fp32=f32))
            f = L_.fc1(x, act="gelu")
            x = L_.final_layer_norm(L_.fc2(f, residual=x, res_scale=alpha, out_fp32=f32))
        return x.view(B, n, E), padding_mask


def __getattr__(name):
    """`SegModule` lives in seg_module.py but is reachable as crab_amd.multimodal_encoder.SegModule like in the reference."""
    if name == "SegModule":
        from .seg_module import SegModule
        return SegModule
    raise AttributeError(name)


class AudioEncoder(nn.Module):
    """multimodal_encoder.py:148-186.  `ckpt_path` may be None (weights arrive via load_state_dict) or a BEATs
    checkpoint ({'cfg','model'}) loaded like the reference does (:157-161)."""

    def __init__(self, ckpt_path=None, cfg: Optional[Dict] = None, device="cuda"):
        super().__init__()
        model_sd = None
        if ckpt_path is not None:
            ck = torch.load(ckpt_path, map_location='cpu')
            cfg, model_sd = ck['cfg'], ck['model']
        bc = BEATsConfig(cfg)
        bc.encoder_layerdrop = 0.
        self.audio_encoder = BEATs(bc, device=device)
        if model_sd is not None:
            self.audio_encoder.load_state_dict(model_sd, strict=False)       # copy_ casts to each parameter's storage (LayerNorm: fp32)

    @torch.no_grad()
    def encode_audio(self, audio):
        emb, _ = self.audio_encoder.extract_features(audio, padding_mask=None, feature_only=True)
        return emb

    def forward(self, audio):
        if audio.dim() == 4:
            b, t, L, d = audio.shape
            e = self.encode_audio(audio.reshape(b * t, L, d))
            return e.reshape(b, t, e.shape[1], e.shape[2])
        return self.encode_audio(audio)
