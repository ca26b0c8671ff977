"""SegModule (the `generate_avs` pixel path) on the HIP kernels -- mirror of reference
`models/multimodal_encoder.py`: SegModule (:268-543), MaskDecoderMultiScale (:891-1143), TwoWayTransformer /
TwoWayAttentionBlock / Attention (:1163-1390), QueryGenerator (:1396-1444), LayerNorm2d (:606-618),
PositionEmbeddingRandom (:795-844), MLP / MLP_conv (:850-888).  Parameter names match the reference state dict
(`seg_module.*` in the AVS `finetune_weights.bin`), including the two `positional_encoding_gaussian_matrix` buffers
that the reference never saves (explicit weight inputs here).

Layout: every feature map is token-major `[h*w, C]` bf16, so Conv1x1 = GEMM, LayerNorm2d = LayerNorm(eps 1e-6),
Conv3x3 = im2col + GEMM, ConvTranspose2d(2,2) = GEMM + pixel shuffle, and all attention (8 heads of 32, or 16 padded
to 32 with zero weight rows) runs on the flash-attention kernel.  Reference quirks reproduced: the QueryGenerator feeds
the original queries to every layer (only the last layer matters, A.4); `multiseg_scalar` / `multiscale_scalar` are the
constants 1/3 and 1/2 (A.5); `num_classes` is chosen per sample from the task name (A.6).
"""
from __future__ import annotations

import math

import torch
from torch import nn

from . import ops
from .multimodal_encoder import LayerNormP, LinearP, _p

BF16 = torch.bfloat16


class _LayerNorm2d(LayerNormP):
    def __init__(self, c, device):
        super().__init__(c, 1e-6, device)


class _Conv(nn.Module):
    """nn.Conv2d parameter container (weight [out,in,k,k], optional bias)."""

    def __init__(self, cin, cout, k, device, bias=True):
        super().__init__()
        self.weight = _p(None, device, cout, cin, k, k)
        if bias:
            self.bias = _p(None, device, cout)
        else:
            self.register_parameter("bias", None)


class _ConvT(nn.Module):
    """nn.ConvTranspose2d(k=2,s=2) parameter container (weight [in,out,2,2], bias [out])."""

    def __init__(self, cin, cout, device):
        super().__init__()
        self.weight = _p(None, device, cin, cout, 2, 2)
        self.bias = _p(None, device, cout)


class PositionEmbeddingRandom(nn.Module):
    def __init__(self, num_pos_feats, device):
        super().__init__()
        # kept in fp32: the phases 2*pi*(c @ G) lose ~0.07 rad when G is rounded to bf16
        self.register_buffer("positional_encoding_gaussian_matrix", torch.zeros(2, num_pos_feats, device=device, dtype=torch.float32))

    def forward(self, size):
        return ops.dense_pe(self.positional_encoding_gaussian_matrix, size[0], size[1])        # token-major [h*w, C]


class Attention(nn.Module):
    """multimodal_encoder.py:1333-1390."""

    def __init__(self, embedding_dim, num_heads, downsample_rate=1, device="cuda"):
        super().__init__()
        self.embedding_dim, self.num_heads = embedding_dim, num_heads
        self.internal_dim = embedding_dim // downsample_rate
        self.q_proj = LinearP(embedding_dim, self.internal_dim, device)
        self.k_proj = LinearP(embedding_dim, self.internal_dim, device)
        self.v_proj = LinearP(embedding_dim, self.internal_dim, device)
        self.out_proj = LinearP(self.internal_dim, embedding_dim, device)
        self._packed = None

    def _pack(self):
        """Heads of 16 are zero-padded to the kernel's minimum head dim of 32 (scores and outputs are unchanged)."""
        key = tuple(p._version for p in self.parameters())
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]
        H, d = self.num_heads, self.internal_dim // self.num_heads
        if d == 32:
            pk = (self.q_proj.weight, self.q_proj.bias, self.k_proj.weight, self.k_proj.bias, self.v_proj.weight, self.v_proj.bias,
                  self.out_proj.weight, 32)
        elif d == 16:
            def pad_rows(w, b):
                E = w.shape[1]
                wp = torch.zeros((H, 32, E), device=w.device, dtype=BF16)
                wp[:, :16] = w.view(H, 16, E)
                bp = torch.zeros((H, 32), device=w.device, dtype=BF16)
                bp[:, :16] = b.view(H, 16)
                return wp.reshape(H * 32, E), bp.reshape(H * 32)
            qw, qb = pad_rows(self.q_proj.weight, self.q_proj.bias)
            kw, kb = pad_rows(self.k_proj.weight, self.k_proj.bias)
            vw, vb = pad_rows(self.v_proj.weight, self.v_proj.bias)
            ow = torch.zeros((self.embedding_dim, H, 32), device=qw.device, dtype=BF16)
            ow[:, :, :16] = self.out_proj.weight.view(self.embedding_dim, H, 16)
            pk = (qw, qb, kw, kb, vw, vb, ow.reshape(self.embedding_dim, H * 32), 32)
        else:
            raise NotImplementedError(f"attention head dim {d} (supported: 16, 32)")
        self._packed = (key, pk)
        return pk

    def forward(self, q, k, v, residual=None, B=1):
        """q [B*Sq,E], k/v [B*Skv,E] token-major, the B samples stacked -> out_proj(attn) (+ residual)."""
        qw, qb, kw, kb, vw, vb, ow, dp = self._pack()
        H = self.num_heads
        Sq, Skv = q.shape[0] // B, k.shape[0] // B
        ip = H * dp
        qp = ops.gemm(q, qw, bias=qb)
        kv = torch.empty((B * Skv, 2 * ip), device=q.device, dtype=BF16)
        ops.gemm(k, kw, bias=kb, out=kv[:, :ip])
        ops.gemm(v, vw, bias=vb, out=kv[:, ip:])
        att = _attend(qp, kv, B, H, Sq, Skv, dp, 1.0 / math.sqrt(self.internal_dim // H))
        return ops.gemm(att, ow, bias=self.out_proj.bias, residual=residual)


def _attend(qp, kv, B, H, Sq, Skv, d, scale):
    """qp [B*Sq, H*d]; kv [B*Skv, 2*H*d] = [k | v] -> attention output [B*Sq, H*d]."""
    dev = qp.device
    Sp = (Skv + 7) // 8 * 8
    vt = torch.zeros((B, H, d, Sp), device=dev, dtype=BF16)
    ops.qkv_rope_split(kv, None, None, None, vt, B, Skv, 0, H, d, 1, 0, None)
    out = torch.empty((B * Sq, H * d), device=dev, dtype=BF16)
    ldk = kv.stride(0)
    ops.attn_fwd(qp, kv, vt, out, q_strides=(Sq * H * d, d, H * d), k_strides=(Skv * ldk, d, ldk),
                 vt_strides=(H * d * Sp, d * Sp, Sp), o_strides=(Sq * H * d, H * d), B=B, H=H, Hk=H, Sq=Sq, Skv=Skv,
                 head_dim=d, scale=scale)
    return out


class _MLPBlock(nn.Module):
    def __init__(self, dim, mlp_dim, device):
        super().__init__()
        self.lin1 = LinearP(dim, mlp_dim, device)
        self.lin2 = LinearP(mlp_dim, dim, device)


class TwoWayAttentionBlock(nn.Module):
    """multimodal_encoder.py:1257-1330."""

    def __init__(self, dim, heads, mlp_dim, down, skip_first_layer_pe, device):
        super().__init__()
        self.self_attn = Attention(dim, heads, device=device)
        self.norm1 = LayerNormP(dim, 1e-5, device)
        self.cross_attn_token_to_image = Attention(dim, heads, down, device=device)
        self.norm2 = LayerNormP(dim, 1e-5, device)
        self.mlp = _MLPBlock(dim, mlp_dim, device)
        self.norm3 = LayerNormP(dim, 1e-5, device)
        self.norm4 = LayerNormP(dim, 1e-5, device)
        self.cross_attn_image_to_token = Attention(dim, heads, down, device=device)
        self.skip_first_layer_pe = skip_first_layer_pe

    def forward(self, queries, keys, query_pe, key_pe, B=1):
        """queries / query_pe [B*Nq, C], keys [B*hw, C] (samples stacked), key_pe [hw, C] shared by the samples (add_rows broadcasts it)."""
        if self.skip_first_layer_pe:
            queries = self.self_attn(queries, queries, queries, B=B)
        else:
            q = ops.add_rows(queries, query_pe)
            queries = self.self_attn(q, q, queries, residual=queries, B=B)
        queries = self.norm1(queries)
        q, k = ops.add_rows(queries, query_pe), ops.add_rows(keys, key_pe)
        queries = self.norm2(self.cross_attn_token_to_image(q, k, keys, residual=queries, B=B))
        m = self.mlp.lin2(self.mlp.lin1(queries, act="relu"), residual=queries)
        queries = self.norm3(m)
        q = ops.add_rows(queries, query_pe)
        keys = self.norm4(self.cross_attn_image_to_token(k, q, queries, residual=keys, B=B))
        return queries, keys


class TwoWayTransformer(nn.Module):
    """multimodal_encoder.py:1163-1254."""

    def __init__(self, depth, embedding_dim, num_heads, mlp_dim, attention_downsample_rate=2, device="cuda"):
        super().__init__()
        self.layers = nn.ModuleList([TwoWayAttentionBlock(embedding_dim, num_heads, mlp_dim, attention_downsample_rate, i == 0, device)
                                     for i in range(depth)])
        self.final_attn_token_to_image = Attention(embedding_dim, num_heads, attention_downsample_rate, device=device)
        self.norm_final_attn = LayerNormP(embedding_dim, 1e-5, device)

    def forward(self, keys, key_pe, point_embedding, B=1):
        """keys [B*h*w, C] (= image_embedding flattened, samples stacked), key_pe [h*w, C], point_embedding [B*Nq, C]."""
        queries = point_embedding
        for layer in self.layers:
            queries, keys = layer(queries, keys, point_embedding, key_pe, B=B)
        q, k = ops.add_rows(queries, point_embedding), ops.add_rows(keys, key_pe)
        queries = self.norm_final_attn(self.final_attn_token_to_image(q, k, keys, residual=queries, B=B))
        return queries, keys


class _TorchMHA(nn.Module):
    """nn.MultiheadAttention(batch_first=True) parameter container (packed in_proj)."""

    def __init__(self, E, heads, device):
        super().__init__()
        self.embed_dim, self.num_heads = E, heads
        self.in_proj_weight = _p(None, device, 3 * E, E)
        self.in_proj_bias = _p(None, device, 3 * E)
        self.out_proj = LinearP(E, E, device)

    def forward(self, q, kv, residual, B=1):
        E, H = self.embed_dim, self.num_heads
        qp = ops.gemm(q, self.in_proj_weight[:E], bias=self.in_proj_bias[:E])
        kvp = ops.gemm(kv, self.in_proj_weight[E:], bias=self.in_proj_bias[E:])                # [B*Skv, 2E] = [k | v]
        att = _attend(qp, kvp, B, H, q.shape[0] // B, kv.shape[0] // B, E // H, 1.0 / math.sqrt(E // H))
        return self.out_proj(att, residual=residual)


class _QGLayer(nn.Module):
    def __init__(self, E, heads, hidden, device):
        super().__init__()
        self.self_attn = _TorchMHA(E, heads, device)
        self.cross_attn = _TorchMHA(E, heads, device)
        self.ffn = nn.Sequential(LinearP(E, hidden, device), nn.Identity(), LinearP(hidden, E, device))
        self.norm1 = LayerNormP(E, 1e-5, device)
        self.norm2 = LayerNormP(E, 1e-5, device)
        self.norm3 = LayerNormP(E, 1e-5, device)

    def forward(self, query, feat, B=1):
        """query [Nq, E] (the learned queries: the same for every sample), feat [B, E] = one sparse prompt row per sample -> [B*Nq, E].
        The self-attention does not see the sample, so it runs once; its output is replicated (a device copy) in front of the cross-attention."""
        query = self.norm1(self.self_attn(query, query, residual=query))
        if B > 1:
            Nq, E = query.shape
            rep = torch.empty((B * Nq, E), device=query.device, dtype=BF16)
            ops.copy_rows_batched(query, E, 0, rep, E, Nq * E, B, Nq, E)
            query = rep
        query = self.norm2(self.cross_attn(query, feat, residual=query, B=B))
        f = self.ffn[2](self.ffn[0](query, act="gelu"), residual=query)
        return self.norm3(f)


class QueryGenerator(nn.Module):
    """multimodal_encoder.py:1422-1444: every layer consumes the ORIGINAL queries; the last layer's output is returned."""

    def __init__(self, num_layers, embed_dim=256, num_heads=8, hidden_dim=1024, device="cuda"):
        super().__init__()
        self.layers = nn.ModuleList([_QGLayer(embed_dim, num_heads, hidden_dim, device) for _ in range(num_layers)])

    def forward(self, avs_query, sparse_embedding, B=1):
        return self.layers[-1](avs_query, sparse_embedding, B=B)    # earlier layers' outputs are discarded by the reference


class _MLP(nn.Module):
    def __init__(self, dims, device, conv=False):
        super().__init__()
        mk = (lambda i, o: _Conv(i, o, 1, device)) if conv else (lambda i, o: LinearP(i, o, device))
        self.layers = nn.ModuleList([mk(i, o) for i, o in zip(dims[:-1], dims[1:])])


class MaskDecoderMultiScale(nn.Module):
    """multimodal_encoder.py:891-1143."""

    def __init__(self, transformer_dim, depth, image_feature_scale_num, avs_query_num, query_generator_num_layers, device):
        super().__init__()
        D = transformer_dim
        self.transformer_dim, self.avs_query_num = D, avs_query_num
        self.transformer = nn.ModuleList([TwoWayTransformer(depth, D, 8, 2048, device=device) for _ in range(image_feature_scale_num)])
        self.avs_query_tokens = nn.Module()
        self.avs_query_tokens.weight = _p(None, device, avs_query_num, D)
        self.query_generator = QueryGenerator(query_generator_num_layers, D, 8, 2048, device=device)
        self.hyper_mlp_out = _MLP([avs_query_num, D, D, D // 8], device, conv=True)
        self.hyper_mlp = _MLP([D, D, D, D // 8], device)
        self.output_upscaling = nn.Sequential(_ConvT(D, D // 8, device), _LayerNorm2d(D // 8, device), nn.Identity())
        self.upsample_2x = nn.Sequential(_ConvT(D, D, device), _LayerNorm2d(D, device), nn.Identity())
        self.pe1 = PositionEmbeddingRandom(D // 2, device)
        self.level_embed = nn.Module()
        self.level_embed.weight = _p(None, device, image_feature_scale_num, D)
        self.ms3_s4_classfier = _Conv(D // 8, 1, 1, device, bias=False)              # (sic)
        self.avss_classifier = _Conv(D // 8, 71, 1, device, bias=False)
        self._packed = None

    def _pack(self):
        key = tuple(p._version for p in self.parameters())
        if self._packed is not None and self._packed[0] == key:
            return self._packed[1]

        def convt(m):           # [Cin, Co, 2, 2] -> [(dy,dx,co), Cin]
            w = m.weight
            return w.permute(2, 3, 1, 0).reshape(4 * w.shape[1], w.shape[0]).contiguous()

        nq = self.avs_query_num
        nqp = (nq + 7) // 8 * 8
        w0 = self.hyper_mlp_out.layers[0].weight.reshape(-1, nq)
        w0p = torch.zeros((w0.shape[0], nqp), device=w0.device, dtype=BF16)
        w0p[:, :nq] = w0
        pk = dict(up2=convt(self.upsample_2x[0]), ups=convt(self.output_upscaling[0]), nqp=nqp, w0p=w0p,
                  w1=self.hyper_mlp_out.layers[1].weight.reshape(self.hyper_mlp_out.layers[1].weight.shape[0], -1),
                  w2=self.hyper_mlp_out.layers[2].weight.reshape(self.hyper_mlp_out.layers[2].weight.shape[0], -1),
                  cls1=self.ms3_s4_classfier.weight.reshape(1, -1), cls71=self.avss_classifier.weight.reshape(71, -1))
        self._packed = (key, pk)
        return pk

    def predict_masks(self, img, image_pe, sparse, no_mask, level, prev, task_name, h, w, B=1):
        """B samples of one task stacked sample-major: img [B*h*w, D] token-major, sparse [B, D] (row b = sample b's prompt of this level; any
        row stride), prev [B*(h'*w'), ncls] masks of the previous level (or None) -> (masks [B*(2h')(2w'), ncls] token-major, h_out, w_out).
        Every step is row-wise (GEMMs, norms, gates), an attention with a batch dimension, or a 2x pixel shuffle - which sees the stacked samples
        as ONE image of B*h rows (the shuffle never mixes rows of different samples) - except the query x pixel product, a batched GEMM."""
        pk = self._pack()
        D, nq = self.transformer_dim, self.avs_query_num
        tokens = self.query_generator(self.avs_query_tokens.weight, sparse, B=B)
        tokens = ops.add_rows(tokens, self.level_embed.weight[level:level + 1])
        src = img
        if level > 0:
            g = ops.gemm(src, pk["up2"])
            src = ops.pixel_shuffle2x(g, self.upsample_2x[0].bias, B * h, w, D)
            h, w = 2 * h, 2 * w
            src = ops.act_inplace(self.upsample_2x[1](src), "gelu")
            ops.mask_gate(prev, src)
            image_pe = self.pe1((h, w))
        src = ops.add_rows(src, no_mask)          # dense prompt = no_mask_embed broadcast (its bilinear resize is the identity)
        hs, keys = self.transformer[level](src, image_pe, tokens, B=B)
        assert hs.shape[0] == B * nq
        t = hs
        lay = self.hyper_mlp.layers
        t = lay[2](lay[1](lay[0](t, act="relu"), act="relu"))                               # [B*nq, D/8]
        g = ops.gemm(keys, pk["ups"])
        up = ops.pixel_shuffle2x(g, self.output_upscaling[0].bias, B * h, w, D // 8)
        h2, w2 = 2 * h, 2 * w
        up = ops.act_inplace(self.output_upscaling[1](up), "gelu")                          # [B*h2*w2, D/8]
        masks = torch.zeros((B * h2 * w2, pk["nqp"]), device=up.device, dtype=BF16)         # K of the next GEMM padded to 8
        if B == 1:
            ops.gemm(up, t, out=masks[:, :nq])                                              # masks[pix, q] = <up[pix], t[q]>
        else:                                                                               # the same product per sample: one batched launch
            gd = ops.GemmDesc()
            gd.A, gd.B, gd.C = up.data_ptr(), t.data_ptr(), masks.data_ptr()
            gd.lda, gd.ldb, gd.ldc = up.stride(0), t.stride(0), masks.stride(0)
            gd.M, gd.N, gd.K = h2 * w2, nq, up.shape[1]
            gd.res_scale, gd.batch, gd.nb0 = 1.0, B, B
            gd.sA0, gd.sB0, gd.sC0 = h2 * w2 * up.stride(0), nq * t.stride(0), h2 * w2 * masks.stride(0)
            ops.gemm_desc(gd, up.device.index or 0)
        mo = self.hyper_mlp_out.layers
        x = ops.gemm(masks, pk["w0p"], bias=mo[0].bias, act="relu")
        x = ops.gemm(x, pk["w1"], bias=mo[1].bias, act="relu")
        x = ops.gemm(x, pk["w2"], bias=mo[2].bias)
        pred = ops.gemm(x, pk["cls71"] if task_name == 'avss' else pk["cls1"])
        return pred, h2, w2


class SegModule(nn.Module):
    """multimodal_encoder.py:268-543 (inference branch; the loss branch :450-497 is training-only)."""

    def __init__(self, d_model=3584, vit_image_embedding_dim=1024, prompt_embed_dim=256, image_scale_nums=2,
                 mask_decoder_transformer_depth=2, token_nums_per_scale=3, avs_query_num=300, num_classes=1,
                 query_generator_num_layers=2, image_size=224, patch_size=14, image_embedding_size=16, dice_loss_weight=0.5,
                 bce_loss_weight=2.0, device="cuda"):
        super().__init__()
        assert patch_size * image_embedding_size == image_size
        self.image_scale_nums, self.token_nums_per_scale = image_scale_nums, token_nums_per_scale
        self.image_embedding_size, self.image_size, self.patch_size = image_embedding_size, image_size, patch_size
        self.num_classes = num_classes
        P = prompt_embed_dim
        self.text_hidden_fcs = nn.ModuleList([nn.Sequential(LinearP(d_model, d_model, device), nn.Identity(),
                                                            LinearP(d_model, P, device), nn.Identity())])
        self.no_mask_embed = nn.Module()
        self.no_mask_embed.weight = _p(None, device, 1, P)
        self.image_feature_neck = nn.Sequential(_Conv(vit_image_embedding_dim, P, 1, device, bias=False), _LayerNorm2d(P, device),
                                                _Conv(P, P, 3, device, bias=False), _LayerNorm2d(P, device))
        self.pe_layer = PositionEmbeddingRandom(P // 2, device)
        self.mask_decoder = MaskDecoderMultiScale(P, mask_decoder_transformer_depth, image_scale_nums, avs_query_num,
                                                  query_generator_num_layers, device)
        self._packed = None

    def get_dense_pe(self):
        return self.pe_layer((self.image_embedding_size, self.image_embedding_size))

    def _pack(self):
        n = self.image_feature_neck
        key = (n[0].weight._version, n[2].weight._version)
        if self._packed is None or self._packed[0] != key:
            w1 = n[0].weight.reshape(n[0].weight.shape[0], -1)
            w3 = n[2].weight.permute(0, 2, 3, 1).reshape(n[2].weight.shape[0], -1).contiguous()   # [Co, (ky,kx,ci)]
            self._packed = (key, (w1, w3))
        return self._packed[1]

    # samples of one class count that go through the mask decoder together (the final masks are fp32 [B, ncls, 224, 224]: 14 MB per avss sample)
    BATCH_BINARY, BATCH_AVSS = 128, 32

    @torch.no_grad()
    def forward(self, pred_embeddings, multi_scale_image_feature_list, low_res_mask_size=112, gt_mask=None, batch_task_names=[]):
        """multimodal_encoder.py:368-448 (inference branch).  The reference walks the samples one by one (:381); here the samples that share a
        class count (`71 if 'avss' else 1`, :419) run through the neck and both mask-decoder levels TOGETHER (rows stacked sample-major, attention
        with a batch dimension) - a sample's result does not depend on its companions (no step mixes rows of different samples).  Returns the
        reference's structure: {'pred_masks': [per sample fp32 [num_classes, 224, 224]]}."""
        if gt_mask is not None:
            raise NotImplementedError("mask losses are training-only (multimodal_encoder.py:450-497)")
        dev = self.no_mask_embed.weight.device
        pe_in = pred_embeddings.to(device=dev, dtype=BF16)
        bs, n, dm = pe_in.shape
        fcs = self.text_hidden_fcs[0]
        e = fcs[2](fcs[0](pe_in.reshape(bs * n, dm), act="relu"))                             # [bs*n, P]
        pred_masks = [None] * bs
        groups = {}
        for i in range(bs):
            groups.setdefault(71 if batch_task_names[i] == 'avss' else 1, []).append(i)
        for ncls, idx in groups.items():
            cap = self.BATCH_AVSS if ncls > 1 else self.BATCH_BINARY
            for c0 in range(0, len(idx), cap):
                part = idx[c0:c0 + cap]
                out = self._forward_group(e, n, part, multi_scale_image_feature_list, low_res_mask_size, ncls, batch_task_names[part[0]])
                for j, i in enumerate(part):
                    pred_masks[i] = out[j]
        return {'pred_masks': pred_masks}

    def _forward_group(self, e, n, idx, feats_list, low_res, ncls, task):
        """The samples `idx` (one class count) through neck + mask decoder + the two bilinear resizes -> fp32 [len(idx), ncls, 224, 224]."""
        S, T, es = self.image_scale_nums, self.token_nums_per_scale, self.image_embedding_size
        dev = e.device
        B = len(idx)
        hw = es * es
        contiguous_run = idx == list(range(idx[0], idx[0] + B))
        sel = None if contiguous_run else torch.tensor(idx, device=dev)
        # sparse prompt per (sample, scale): sum_k (1/T) e[sample, scale, k]  (multiseg_scalar = 1/T constants, appendix A.5; obj_nums == 1)
        eg = e.view(-1, n, e.shape[1])[idx[0]:idx[0] + B] if contiguous_run else e.view(-1, n, e.shape[1])[sel]
        sparse = ops.group_mean(eg.reshape(B * n, -1), B * S, T, 1.0 / T).view(B, S, -1)       # [B, S, P]
        # neck over both levels of every sample at once, level-major so that each level is one contiguous [B*hw, P] slab:
        # conv1x1 -> LN2d -> conv3x3 -> LN2d
        fl = []
        for f in feats_list[:S]:
            f = f.to(device=dev, dtype=BF16)
            fl.append((f[idx[0]:idx[0] + B] if contiguous_run else f[sel])[:, :hw])
        feats = torch.cat(fl, 0).reshape(S * B * hw, -1) if B > 1 or S > 1 else fl[0].reshape(hw, -1)
        w1, w3 = self._pack()
        x = self.image_feature_neck[1](ops.gemm(feats, w1))
        x = self.image_feature_neck[3](ops.gemm(ops.im2col3x3(x, S * B, es, es), w3))         # [S*B*hw, P]
        pe = self.get_dense_pe()
        low = torch.empty((B, ncls, low_res, low_res), device=dev, dtype=torch.float32)
        prev = None
        for l in range(S):
            prev, h2, w2 = self.mask_decoder.predict_masks(x[l * B * hw:(l + 1) * B * hw], pe, sparse[:, l], self.no_mask_embed.weight, l, prev,
                                                           task, es, es, B=B)
            if ncls == 1:                       # one plane per sample: the B planes are the "channels" of ONE resize launch
                ops.bilinear(prev, (h2 * w2, w2, 1), B, h2, w2, low.view(B, low_res, low_res), alpha=1.0 / S, beta=1.0 if l else 0.0)
            else:
                pv = prev.view(B, h2 * w2, ncls)
                for b in range(B):
                    ops.bilinear(pv[b], (1, w2 * ncls, ncls), ncls, h2, w2, low[b], alpha=1.0 / S, beta=1.0 if l else 0.0)
        out = torch.empty((B, ncls, self.image_size, self.image_size), device=dev, dtype=torch.float32)
        ops.bilinear(low, (low_res * low_res, low_res, 1), B * ncls, low_res, low_res, out.view(B * ncls, self.image_size, self.image_size))
        return out
