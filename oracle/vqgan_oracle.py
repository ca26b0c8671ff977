"""CPU restatement of the VQGAN mask tokenizer -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench
cpu_baseline); the product path is crab_amd/vqgan.py -> libcrab_hip.so.

Follows /root/reference: models/multimodal_encoder.py:546-601 (MaskEncoder), models/taming_transformer/vqgan.py:54-99
(VQModel.encode / decode / decode_code / get_codebook_indices), modules.py:29-35 (swish, GroupNorm(32, eps 1e-6)),
:38-75 (Upsample / Downsample), :78-137 (ResnetBlock), :140-192 (AttnBlock), :342-433 (Encoder), :436-538 (Decoder),
quantize.py:272-330 (VectorQuantizer2.forward / get_codebook_entry).  Pinned by tests/golden/vqgan_tiny.npz (outputs of
the reference classes, make_golden.py vqgan)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Sequence

import torch
import torch.nn.functional as F


@dataclass
class VQConfig:
    ch: int = 128
    ch_mult: Sequence[int] = (1, 1, 2, 2, 4)
    num_res_blocks: int = 2
    attn_resolutions: Sequence[int] = (16,)
    resolution: int = 256
    in_channels: int = 3
    out_ch: int = 3
    z_channels: int = 256
    n_embed: int = 16384
    embed_dim: int = 256
    groups: int = 32


# Emulation modes (as oracle/crab_oracle.py has them for the transformer stacks): what a bf16-MFMA implementation of this path rounds.
#   None              : fp32 (the reference)
#   "floor"           : the bf16-OPERAND FLOOR - only what a matrix instruction consumes is rounded, once: conv / 1x1 weights and input maps, the
#                       attention operands q, k, v and the softmax probabilities as the P.V operand.  Everything else (residual stream, GroupNorm
#                       parameters, biases, statistics) stays fp32.  No bf16-MFMA implementation can be closer to the fp32 reference.
#   "storage"         : additionally the HIP path's storage points: the residual stream x between blocks is stored in bf16
#   "storage_fp32res" : the storage emulation with the residual stream kept in fp32 (what an fp32-stream implementation would store)
EMULATE = None


class emulate:
    """`with emulate("floor"):` run the oracle in that emulation mode."""

    def __init__(self, mode):
        assert mode in (None, "floor", "storage", "storage_fp32res")
        self.mode = mode

    def __enter__(self):
        global EMULATE
        self.prev, EMULATE = EMULATE, self.mode

    def __exit__(self, *a):
        global EMULATE
        EMULATE = self.prev


def _op(x):
    """a matrix operand at its point of consumption"""
    return x if EMULATE is None else x.to(torch.bfloat16).float()


def _res(x):
    """the residual stream between blocks: a storage point of the all-bf16 form"""
    return x.to(torch.bfloat16).float() if EMULATE == "storage" else x


def _gn(x, W, pre, groups):
    return F.group_norm(x, groups, W[pre + ".weight"].float(), W[pre + ".bias"].float(), 1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv(x, W, pre, stride=1, padding=1):
    return F.conv2d(_op(x), _op(W[pre + ".weight"].float()), W[pre + ".bias"].float(), stride=stride, padding=padding)


def resnet_block(x, W, pre, cin, cout, groups):
    """modules.py:117-137 (temb is None)."""
    h = _conv(_swish(_gn(x, W, pre + ".norm1", groups)), W, pre + ".conv1")
    h = _conv(_swish(_gn(h, W, pre + ".norm2", groups)), W, pre + ".conv2")
    if cin != cout:
        x = _conv(x, W, pre + ".nin_shortcut", padding=0)
    return _res(x + h)


def attn_block(x, W, pre, groups):
    """modules.py:168-192: single-head attention over h*w positions with C channels."""
    h_ = _gn(x, W, pre + ".norm", groups)
    q, k, v = (_conv(h_, W, pre + "." + n, padding=0) for n in ("q", "k", "v"))
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.softmax(torch.bmm(_op(q), _op(k)) * (int(c) ** (-0.5)), dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(_op(v), _op(w_).permute(0, 2, 1)).reshape(b, c, h, w)
    return _res(x + _conv(h_, W, pre + ".proj_out", padding=0))


def encoder(x, W: Dict[str, torch.Tensor], cfg: VQConfig, pre="encoder"):
    """modules.py:406-433."""
    nres = len(cfg.ch_mult)
    in_mult = (1,) + tuple(cfg.ch_mult)
    h = _res(_conv(x.float(), W, pre + ".conv_in"))
    res = cfg.resolution
    for i in range(nres):
        cin, cout = cfg.ch * in_mult[i], cfg.ch * cfg.ch_mult[i]
        for j in range(cfg.num_res_blocks):
            h = resnet_block(h, W, f"{pre}.down.{i}.block.{j}", cin, cout, cfg.groups)
            cin = cout
            if res in cfg.attn_resolutions:
                h = attn_block(h, W, f"{pre}.down.{i}.attn.{j}", cfg.groups)
        if i != nres - 1:
            h = _res(_conv(F.pad(h, (0, 1, 0, 1)), W, f"{pre}.down.{i}.downsample.conv", stride=2, padding=0))
            res //= 2
    c = cfg.ch * cfg.ch_mult[-1]
    h = resnet_block(h, W, pre + ".mid.block_1", c, c, cfg.groups)
    h = attn_block(h, W, pre + ".mid.attn_1", cfg.groups)
    h = resnet_block(h, W, pre + ".mid.block_2", c, c, cfg.groups)
    return _conv(_swish(_gn(h, W, pre + ".norm_out", cfg.groups)), W, pre + ".conv_out")


def decoder(z, W: Dict[str, torch.Tensor], cfg: VQConfig, pre="decoder"):
    """modules.py:506-538."""
    nres = len(cfg.ch_mult)
    c = cfg.ch * cfg.ch_mult[-1]
    res = cfg.resolution // 2 ** (nres - 1)
    h = _res(_conv(z.float(), W, pre + ".conv_in"))
    h = resnet_block(h, W, pre + ".mid.block_1", c, c, cfg.groups)
    h = attn_block(h, W, pre + ".mid.attn_1", cfg.groups)
    h = resnet_block(h, W, pre + ".mid.block_2", c, c, cfg.groups)
    cin = c
    for i in reversed(range(nres)):
        cout = cfg.ch * cfg.ch_mult[i]
        for j in range(cfg.num_res_blocks + 1):
            h = resnet_block(h, W, f"{pre}.up.{i}.block.{j}", cin, cout, cfg.groups)
            cin = cout
            if res in cfg.attn_resolutions:
                h = attn_block(h, W, f"{pre}.up.{i}.attn.{j}", cfg.groups)
        if i != 0:
            h = _res(_conv(F.interpolate(h, scale_factor=2.0, mode="nearest"), W, f"{pre}.up.{i}.upsample.conv"))
            res *= 2
    return _conv(_swish(_gn(h, W, pre + ".norm_out", cfg.groups)), W, pre + ".conv_out")


def quantize_indices(h, W, pre="quantize"):
    """quantize.py:281-290: nearest codebook entry of every position, flattened (b, h, w) order."""
    e = W[pre + ".embedding.weight"].float()
    z = h.permute(0, 2, 3, 1).reshape(-1, e.shape[1])
    d = (z ** 2).sum(1, keepdim=True) + (e ** 2).sum(1) - 2 * z @ e.t()
    return torch.argmin(d, dim=1)


def get_codebook_indices(x, W, cfg: VQConfig):
    """vqgan.py:93-98: [b,3,H,W] -> [b, (H/16)*(W/16)] codebook ids."""
    h = _conv(encoder(x, W, cfg), W, "quant_conv", padding=0)
    return quantize_indices(h, W).reshape(x.shape[0], -1)


def encode_latents(x, W, cfg: VQConfig):
    """The pre-quantisation latents (encoder + quant_conv), [b, embed_dim, h, w]: what the argmin is taken over."""
    return _conv(encoder(x, W, cfg), W, "quant_conv", padding=0)


def decode_code(code, W, cfg: VQConfig):
    """vqgan.py:69-75: ids [b, n] -> image [b, out_ch, H, W]."""
    bs, n = code.shape
    s = int(n ** 0.5)
    zq = W["quantize.embedding.weight"].float()[code.reshape(-1)].view(bs, s, s, -1).permute(0, 3, 1, 2)
    return decoder(_conv(zq, W, "post_quant_conv", padding=0), W, cfg)


def encode_mask(mask, W, cfg: VQConfig, token_shift: int):
    """multimodal_encoder.py:575-581 (W keys relative to `vqgan.`)."""
    return get_codebook_indices(mask, W, cfg) + token_shift


def decode_mask(indices, W, cfg: VQConfig, token_shift: int):
    """multimodal_encoder.py:584-590."""
    tokens = torch.clip((indices - token_shift).long(), 0, cfg.n_embed - 1)
    return decode_code(tokens, W, cfg)
