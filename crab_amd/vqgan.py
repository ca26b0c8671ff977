"""VQGAN mask tokenizer on the HIP kernels -- mirror of reference `models/multimodal_encoder.py:546-601`
(MaskEncoder) and `models/taming_transformer/{vqgan,modules,quantize}.py` (VQModel, Encoder, Decoder, ResnetBlock,
AttnBlock, Downsample, Upsample, VectorQuantizer2), inference paths only: `encode_mask` (image -> 256 codebook ids +
token_shift) and `decode_mask` (ids -> image).  Parameter names match the reference state dict
(`mask_encoder.vqgan.encoder.down.0.block.0.norm1.weight`, ..., `quantize.embedding.weight`), so the taming
`vqgan_imagenet_f16_16384` checkpoint loads with `load_state_dict`.

Layout: feature maps are token-major `[b*h*w, C]` bf16 (as in seg_module.py).  Conv3x3 = im2col + MFMA GEMM (residual add
fused in the GEMM epilogue), Conv1x1 = GEMM, GroupNorm(32)+swish = one two-pass kernel, Downsample = strided im2col
(the reference's asymmetric (0,1,0,1) padding), Upsample = nearest 2x + conv.  AttnBlock is single-head attention over
h*w = 256 positions with 512 channels: scores and P.V are plain GEMMs around a row-softmax kernel (V is produced
transposed by swapping the GEMM operands; its bias is added after P.V, rows of P sum to one).

The quantiser is INDEX work and runs in fp32 (r06): quant_conv leaves its latents unrounded (fp32 GEMM output), the codebook is
held in fp32 (16 MB at 16384 x 256) and ops.vq_nearest_f32 evaluates the reference's fp32 expression (|z|^2 + |e|^2) - 2 z.e per
entry with vector FMAs, first minimum wins as torch.argmin - given the same latents the ids equal the reference's wherever its
margin exceeds fp32 summation noise.  GroupNorm weights / biases are fp32 as well (not matrix operands; ops.NORM_PARAMS_FP32)."""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import nn

from . import ops
from .multimodal_encoder import _p

BF16 = torch.bfloat16
GROUPS = 32
# the encoder in front of the quantiser runs in its PRECISE form (fp32 activations, split-bf16 MFMA operands: csrc/vq_ops.hip) unless
# CRAB_VQ_PRECISE=0 (A/B: the bf16-operand form, whose latents sit at the bf16 operand floor and flip ids at sub-floor margins)
import os as _os
VQ_PRECISE = _os.environ.get("CRAB_VQ_PRECISE", "1") != "0"


class Conv2d(nn.Module):
    """nn.Conv2d parameter container (weight [out,in,k,k], bias [out]); `packed()` is the GEMM operand [out, k*k*in_pad]
    in (ky, kx, ci) order with the input channels zero-padded to a multiple of 8 (16-byte im2col vectors)."""

    def __init__(self, cin: int, cout: int, k: int, device, dtype=BF16):
        """dtype = storage of weight / bias: bf16 (the decoder: plain MFMA operands) or fp32 (the encoder in front of the quantiser, whose
        precise form splits the fp32 weight into hi + lo bf16 operands; a bf16-rounded weight alone would cost 2^-9 of every product)."""
        super().__init__()
        self.cin, self.cout, self.k = cin, cout, k
        self.weight = _p(None, device, cout, cin, k, k, dtype=dtype)
        self.bias = _p(None, device, cout, dtype=dtype)
        self._pk = None
        self._pk3 = None
        self._b16 = None

    @property
    def cin_pad(self) -> int:
        return (self.cin + 7) // 8 * 8

    def packed(self) -> torch.Tensor:
        key = (self.weight.data_ptr(), self.weight._version)
        if self._pk is None or self._pk[0] != key:
            w = self.weight if self.weight.dtype == BF16 else ops.cast_bf16(self.weight)
            w = w.permute(0, 2, 3, 1)                                          # [Co, ky, kx, Ci]
            if self.cin_pad != self.cin:
                w = torch.nn.functional.pad(w, (0, self.cin_pad - self.cin))
            self._pk = (key, w.reshape(self.cout, -1).contiguous())
        return self._pk[1]

    def bias16(self) -> torch.Tensor:
        """the bf16 GEMM epilogue's bias operand"""
        if self.bias.dtype == BF16:
            return self.bias
        key = (self.bias.data_ptr(), self.bias._version)
        if self._b16 is None or self._b16[0] != key:
            self._b16 = (key, ops.cast_bf16(self.bias))
        return self._b16[1]

    def packed3(self) -> torch.Tensor:
        """The split operand of the precise form: [Co, k*k * 3*cin_pad], per tap [w_hi | w_hi | w_lo] (ops.split3 pattern 1) against the
        activation map's [x_hi | x_lo | x_hi]."""
        key = (self.weight.data_ptr(), self.weight._version)
        if self._pk3 is None or self._pk3[0] != key:
            w = self.weight.float().permute(0, 2, 3, 1).reshape(self.cout * self.k * self.k, self.cin).contiguous()      # layout glue
            self._pk3 = (key, ops.split3(w, 1).reshape(self.cout, -1))
        return self._pk3[1]


class GroupNorm(nn.Module):
    def __init__(self, c: int, device):
        super().__init__()
        dt = torch.float32 if ops.NORM_PARAMS_FP32 else BF16
        self.weight = _p(None, device, c, fill=1.0, dtype=dt)
        self.bias = _p(None, device, c, dtype=dt)

    def __call__(self, x: torch.Tensor, B: int, HW: int, swish: bool) -> torch.Tensor:
        if x.dtype == torch.float32:                          # the precise encoder: fp32 in / out / parameters
            return ops.groupnorm_f32(x, B, HW, GROUPS, self.weight.float(), self.bias.float(), 1e-6, swish)
        return ops.groupnorm(x, B, HW, GROUPS, self.weight, self.bias, 1e-6, swish)


def _conv3x3(x: torch.Tensor, conv: Conv2d, B: int, h: int, w: int, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    if x.dtype == torch.float32:                              # precise form: fp32 map -> split operand -> MFMA GEMM over 3x K -> fp32 (+ fp32 residual)
        y = ops.gemm(ops.im2col3x3(ops.split3(x, 0), B, h, w), conv.packed3(), residual=residual, out_fp32=True)
        return ops.add_bias_f32(y, conv.bias.float())
    return ops.gemm(ops.im2col3x3(x, B, h, w), conv.packed(), bias=conv.bias16(), residual=residual)


def _conv1x1(x: torch.Tensor, conv: Conv2d, residual: Optional[torch.Tensor] = None, out_fp32: bool = False) -> torch.Tensor:
    if x.dtype == torch.float32:
        y = ops.gemm(ops.split3(x, 0), conv.packed3(), residual=residual, out_fp32=True)
        return ops.add_bias_f32(y, conv.bias.float())
    return ops.gemm(x, conv.packed(), bias=conv.bias16(), residual=residual, out_fp32=out_fp32)


class ResnetBlock(nn.Module):
    """modules.py:78-137 (no timestep embedding: temb_channels = 0 in Encoder / Decoder)."""

    def __init__(self, cin: int, cout: int, device, dtype=BF16):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.norm1 = GroupNorm(cin, device)
        self.conv1 = Conv2d(cin, cout, 3, device, dtype)
        self.norm2 = GroupNorm(cout, device)
        self.conv2 = Conv2d(cout, cout, 3, device, dtype)
        if cin != cout:
            self.nin_shortcut = Conv2d(cin, cout, 1, device, dtype)

    def forward(self, x, B, h, w):
        t = _conv3x3(self.norm1(x, B, h * w, True), self.conv1, B, h, w)
        if self.in_channels != self.out_channels:
            x = _conv1x1(x, self.nin_shortcut)
        return _conv3x3(self.norm2(t, B, h * w, True), self.conv2, B, h, w, residual=x)      # x + h fused in the epilogue


class AttnBlock(nn.Module):
    """modules.py:140-192."""

    def __init__(self, c: int, device, dtype=BF16):
        super().__init__()
        self.in_channels = c
        self.norm = GroupNorm(c, device)
        self.q, self.k, self.v, self.proj_out = (Conv2d(c, c, 1, device, dtype) for _ in range(4))

    def _forward_precise(self, x, B, h, w):
        """The same block on fp32 maps with split operands: scores, V^T and P.V are MFMA GEMMs over a tripled K, the softmax stays in fp32."""
        C, HW = self.in_channels, h * w
        hn = self.norm(x, B, HW, False)
        q = _conv1x1(hn, self.q)
        k = _conv1x1(hn, self.k)
        wv3 = ops.split3(self.v.weight.float().reshape(C, C), 0)                  # V^T = Wv . hn^T: the weight is the A operand here
        o = torch.empty((B * HW, C), device=x.device, dtype=torch.float32)
        for b in range(B):
            sl = slice(b * HW, (b + 1) * HW)
            scores = ops.gemm(ops.split3(q[sl], 0), ops.split3(k[sl], 1), out_fp32=True)
            p = ops.softmax_rows_f32(scores, float(int(C) ** (-0.5)))
            vt = ops.gemm(wv3, ops.split3(hn[sl], 1), out_fp32=True)                # [C, HW] (bias added after P.V: rows of P sum to one)
            ops.gemm(ops.split3(p, 0), ops.split3(vt, 1), out=o[sl])
        ops.add_bias_f32(o, self.v.bias.float())
        return _conv1x1(o, self.proj_out, residual=x)

    def forward(self, x, B, h, w):
        if x.dtype == torch.float32:
            return self._forward_precise(x, B, h, w)
        C, HW = self.in_channels, h * w
        hn = self.norm(x, B, HW, False)
        q = _conv1x1(hn, self.q)
        k = _conv1x1(hn, self.k)
        wv = self.v.packed()
        o = torch.empty((B * HW, C), device=x.device, dtype=BF16)
        HWp = (HW + 7) // 8 * 8                                                   # P . V^T contracts over the pixels: the GEMM's K is a multiple of 8, so a map whose
        pad = HWp != HW                                                           # pixel count is not (masks off the 16-pixel grid) runs on zero-padded P / V^T columns
        if pad:
            pbuf = torch.zeros((HW, HWp), device=x.device, dtype=BF16)
            vbuf = torch.zeros((C, HWp), device=x.device, dtype=BF16)
        for b in range(B):
            sl = slice(b * HW, (b + 1) * HW)
            scores = ops.gemm(q[sl], k[sl], out_fp32=True)                        # [HW, HW] = q . k^T
            p = ops.softmax_rows(scores, float(int(C) ** (-0.5)), out=pbuf[:, :HW] if pad else None)
            vt = ops.gemm(wv, hn[sl], out=vbuf[:, :HW] if pad else None)          # V^T [C, HW] (bias added after P.V)
            ops.gemm(pbuf if pad else p, vbuf if pad else vt, bias=self.v.bias16(), out=o[sl])
        return _conv1x1(o, self.proj_out, residual=x)


class Downsample(nn.Module):
    def __init__(self, c: int, device, dtype=BF16):
        super().__init__()
        self.conv = Conv2d(c, c, 3, device, dtype)

    def forward(self, x, B, h, w):
        oh, ow = (h + 1 - 3) // 2 + 1, (w + 1 - 3) // 2 + 1                         # F.pad (0,1,0,1) then k3 s2 p0
        if x.dtype == torch.float32:
            cols = ops.im2col3x3_strided(ops.split3(x, 0), B, h, w, 2, 0, 0, oh, ow)
            return ops.add_bias_f32(ops.gemm(cols, self.conv.packed3(), out_fp32=True), self.conv.bias.float()), oh, ow
        cols = ops.im2col3x3_strided(x, B, h, w, 2, 0, 0, oh, ow)
        return ops.gemm(cols, self.conv.packed(), bias=self.conv.bias16()), oh, ow


class Upsample(nn.Module):
    def __init__(self, c: int, device):
        super().__init__()
        self.conv = Conv2d(c, c, 3, device)

    def forward(self, x, B, h, w):
        return _conv3x3(ops.upsample_nearest2x(x, B, h, w), self.conv, B, 2 * h, 2 * w), 2 * h, 2 * w


class _Level(nn.Module):
    def __init__(self):
        super().__init__()
        self.block = nn.ModuleList()
        self.attn = nn.ModuleList()


class _Mid(nn.Module):
    def __init__(self, c: int, device, dtype=BF16):
        super().__init__()
        self.block_1 = ResnetBlock(c, c, device, dtype)
        self.attn_1 = AttnBlock(c, device, dtype)
        self.block_2 = ResnetBlock(c, c, device, dtype)

    def forward(self, x, B, h, w):
        return self.block_2(self.attn_1(self.block_1(x, B, h, w), B, h, w), B, h, w)


def _to_tokens(img: torch.Tensor, cpad: int) -> torch.Tensor:
    """[B,C,H,W] fp32 / bf16 -> token-major bf16 [B*H*W, cpad] with zero channel padding (permute = layout glue; the cast
    and the padded copy are library launches)."""
    B, C, H, W = img.shape
    src = ops.cast_bf16(img.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous().to(torch.float32) if img.dtype not in (BF16, torch.float32)
                        else img.permute(0, 2, 3, 1).reshape(B * H * W, C).contiguous())
    if cpad == C:
        return src
    t = torch.zeros((B * H * W, cpad), device=img.device, dtype=BF16)
    ops.copy_rows(src, t, B * H * W, C)
    return t


class Encoder(nn.Module):
    """modules.py:342-433."""

    def __init__(self, *, ch, ch_mult, num_res_blocks, attn_resolutions, in_channels, resolution, z_channels, device, **_ignore):
        super().__init__()
        self.ch, self.num_resolutions, self.num_res_blocks, self.resolution = ch, len(ch_mult), num_res_blocks, resolution
        dt = torch.float32 if VQ_PRECISE else BF16          # the encoder's weights stay as the checkpoint holds them when the precise form runs
        self.precise = VQ_PRECISE
        self.conv_in = Conv2d(in_channels, ch, 3, device, dt)
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        curr = resolution
        block_in = ch
        for i in range(self.num_resolutions):
            lvl = _Level()
            block_in, block_out = ch * in_mult[i], ch * ch_mult[i]
            for _ in range(num_res_blocks):
                lvl.block.append(ResnetBlock(block_in, block_out, device, dt))
                block_in = block_out
                if curr in attn_resolutions:
                    lvl.attn.append(AttnBlock(block_in, device, dt))
            if i != self.num_resolutions - 1:
                lvl.downsample = Downsample(block_in, device, dt)
                curr //= 2
            self.down.append(lvl)
        self.mid = _Mid(block_in, device, dt)
        self.norm_out = GroupNorm(block_in, device)
        self.conv_out = Conv2d(block_in, z_channels, 3, device, dt)

    def forward(self, img: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
        """-> (conv_out map [B*h*w, z_channels], h, w): fp32 in the precise form (every step below then takes its fp32 branch), bf16 otherwise."""
        B, _, h, w = img.shape
        if self.precise:
            x0 = img.to(torch.float32).permute(0, 2, 3, 1).reshape(B * h * w, -1).contiguous()       # token-major fp32 (layout glue)
        else:
            x0 = _to_tokens(img, self.conv_in.cin_pad)
        x = _conv3x3(x0, self.conv_in, B, h, w)
        for i, lvl in enumerate(self.down):
            for j, blk in enumerate(lvl.block):
                x = blk(x, B, h, w)
                if len(lvl.attn) > 0:
                    x = lvl.attn[j](x, B, h, w)
            if i != self.num_resolutions - 1:
                x, h, w = lvl.downsample(x, B, h, w)
        x = self.mid(x, B, h, w)
        return _conv3x3(self.norm_out(x, B, h * w, True), self.conv_out, B, h, w), h, w


class Decoder(nn.Module):
    """modules.py:436-538."""

    def __init__(self, *, ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, resolution, z_channels, device, **_ignore):
        super().__init__()
        self.num_resolutions, self.num_res_blocks = len(ch_mult), num_res_blocks
        block_in = ch * ch_mult[-1]
        curr = resolution // 2 ** (self.num_resolutions - 1)
        self.conv_in = Conv2d(z_channels, block_in, 3, device)
        self.mid = _Mid(block_in, device)
        self.up = nn.ModuleList()
        for i in reversed(range(self.num_resolutions)):
            lvl = _Level()
            block_out = ch * ch_mult[i]
            for _ in range(num_res_blocks + 1):
                lvl.block.append(ResnetBlock(block_in, block_out, device))
                block_in = block_out
                if curr in attn_resolutions:
                    lvl.attn.append(AttnBlock(block_in, device))
            if i != 0:
                lvl.upsample = Upsample(block_in, device)
                curr *= 2
            self.up.insert(0, lvl)
        self.norm_out = GroupNorm(block_in, device)
        self.conv_out = Conv2d(block_in, out_ch, 3, device)

    def forward(self, z: torch.Tensor, B: int, h: int, w: int) -> Tuple[torch.Tensor, int, int]:
        x = _conv3x3(z, self.conv_in, B, h, w)
        x = self.mid(x, B, h, w)
        for i in reversed(range(self.num_resolutions)):
            lvl = self.up[i]
            for j, blk in enumerate(lvl.block):
                x = blk(x, B, h, w)
                if len(lvl.attn) > 0:
                    x = lvl.attn[j](x, B, h, w)
            if i != 0:
                x, h, w = lvl.upsample(x, B, h, w)
        return _conv3x3(self.norm_out(x, B, h * w, True), self.conv_out, B, h, w), h, w


class _Embedding(nn.Module):
    def __init__(self, n: int, d: int, device):
        super().__init__()
        self.weight = _p(None, device, n, d, dtype=torch.float32)       # the codebook: fp32, as the checkpoint holds it (the ids are index work)


class VectorQuantizer(nn.Module):
    """quantize.py:213-330 (VectorQuantizer2 without remap): nearest-entry lookup and `get_codebook_entry`."""

    def __init__(self, n_e: int, e_dim: int, device):
        super().__init__()
        self.n_e, self.e_dim = n_e, e_dim
        self.embedding = _Embedding(n_e, e_dim, device)
        self._e2 = None
        self._w16 = None

    def _norms(self) -> torch.Tensor:
        w = self.embedding.weight
        key = (w.data_ptr(), w._version)
        if self._e2 is None or self._e2[0] != key:
            self._e2 = (key, ops.row_sqnorm_f32(w))
        return self._e2[1]

    def indices(self, z: torch.Tensor, offset: int = 0) -> torch.Tensor:
        """z [M, e_dim] fp32 (unrounded latents) -> int64 [M] = offset + argmin_n |z - e_n|^2, every term in fp32."""
        return ops.vq_nearest_f32(z, self.embedding.weight, self._norms(), offset)

    def get_codebook_entry(self, indices: torch.Tensor) -> torch.Tensor:
        """The decode side consumes the entry as a matrix operand (post_quant_conv): a bf16 copy of the codebook, made once per weight version."""
        w = self.embedding.weight
        key = (w.data_ptr(), w._version)
        if self._w16 is None or self._w16[0] != key:
            self._w16 = (key, ops.cast_bf16(w))
        return ops.embedding(indices.reshape(-1).to(torch.int64), self._w16[1])


class VQModel(nn.Module):
    """vqgan.py:9-98, inference methods."""

    def __init__(self, ddconfig: dict, n_embed: int, embed_dim: int, device="cuda", lossconfig=None, ckpt_path=None, **_ignore):
        super().__init__()
        self.embed_dim = embed_dim
        self.encoder = Encoder(**ddconfig, device=device)
        self.decoder = Decoder(**ddconfig, device=device)
        self.quantize = VectorQuantizer(n_embed, embed_dim, device)
        self.quant_conv = Conv2d(ddconfig["z_channels"], embed_dim, 1, device, torch.float32 if VQ_PRECISE else BF16)
        self.post_quant_conv = Conv2d(embed_dim, ddconfig["z_channels"], 1, device)
        if ckpt_path is not None:
            self.load_state_dict(torch.load(ckpt_path, map_location="cpu"), strict=False)     # vqgan.py:42-52

    def encode_latents(self, x: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
        """-> fp32 latents [b*h*w, embed_dim]: the output of quant_conv is what the quantiser measures distances from, so it is not rounded."""
        h, hh, ww = self.encoder(x)
        return _conv1x1(h, self.quant_conv, out_fp32=True), hh, ww

    @torch.no_grad()
    def get_codebook_indices(self, x: torch.Tensor, offset: int = 0) -> torch.Tensor:
        z, _, _ = self.encode_latents(x)
        return self.quantize.indices(z, offset).reshape(x.shape[0], -1)

    @torch.no_grad()
    def decode_code(self, code_b: torch.Tensor) -> torch.Tensor:
        bs, n = code_b.shape
        size = int(math.sqrt(n))
        zq = self.quantize.get_codebook_entry(code_b)
        y, h, w = self.decoder(_conv1x1(zq, self.post_quant_conv), bs, size, size)
        return y.reshape(bs, h, w, -1).permute(0, 3, 1, 2).float()


class MaskEncoder(nn.Module):
    """multimodal_encoder.py:546-601."""

    DDCONFIG = dict(double_z=False, z_channels=256, resolution=256, in_channels=3, out_ch=3, ch=128, ch_mult=(1, 1, 2, 2, 4),
                    num_res_blocks=2, attn_resolutions=(16,), dropout=0.0)

    def __init__(self, token_shift: int = 32000, device="cuda", ddconfig: Optional[dict] = None, n_embed: int = 16384, embed_dim: int = 256,
                 ckpt_path=None):
        super().__init__()
        self.vqgan = VQModel(ddconfig=dict(ddconfig or self.DDCONFIG), n_embed=n_embed, embed_dim=embed_dim, device=device, ckpt_path=ckpt_path)
        self.n_embed = n_embed
        self.token_shift = token_shift

    @torch.no_grad()
    def encode_mask(self, mask: torch.Tensor) -> torch.Tensor:
        """mask [b,c,h,w] -> ids [b, n] + token_shift."""
        dev = self.vqgan.quant_conv.weight.device
        return self.vqgan.get_codebook_indices(mask.to(dev), self.token_shift)

    @torch.no_grad()
    def decode_mask(self, indices: torch.Tensor) -> torch.Tensor:
        """ids [b, n] -> image [b,c,h,w]."""
        dev = self.vqgan.quant_conv.weight.device
        tokens = torch.clip((indices.to(dev) - self.token_shift).to(torch.long), 0, self.n_embed - 1)
        return self.vqgan.decode_code(tokens)

    def forward(self, mask):
        return self.encode_mask(mask)
