"""Deterministic synthetic weights and inputs.

There are no checkpoints offline (Llama-2, CLIP, BEATs and `finetune_weights.bin` are external
downloads, reference README.md:79-89), so every test, golden fixture and benchmark in this repo
runs on seeded random tensors.  The rule below is the single definition of "the weights for
(seed, name, shape)": the golden script loads them into the imported reference, the oracle and the
HIP path regenerate exactly the same tensors on the GPU box, and fixtures carry only the
(name, shape, checksum) table.

Plain torch CPU RNG (plumbing, not product arithmetic).
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Iterable, Sequence, Tuple

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _gen(seed: int, name: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _is_norm_weight(name: str) -> bool:
    n = name.lower()
    leaf = n.rsplit(".", 1)[-1]
    if leaf != "weight":
        return False
    parent = n.rsplit(".", 1)[0]
    keys = ("layernorm", "layer_norm", "_ln", "norm", "layrnorm", ".ln")
    tail = parent.rsplit(".", 1)[-1]
    return any(k in tail for k in keys)


def synth_tensor(name: str, shape: Sequence[int], seed: int = 0, scheme: str = "fan_in",
                 dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """One tensor of the synthetic checkpoint.

    scheme 'fan_in': matrices ~ N(0, 1/fan_in) (keeps activations O(1) in the tiny fixtures so greedy
    margins are comfortable); scheme 'n002': matrices ~ N(0, 0.02^2) (SURVEY.md 8d full-size rule).
    Norm weights are 1 + 0.1 n, biases 0.05 n, so no affine parameter is a silent identity.
    """
    shape = tuple(int(s) for s in shape)
    g = _gen(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    if _is_norm_weight(name) and len(shape) == 1:
        t = 1.0 + 0.1 * torch.randn(shape, generator=g)
    elif leaf == "bias" or (len(shape) == 1 and leaf not in ("weight",)):
        t = 0.05 * torch.randn(shape, generator=g)
    elif leaf == "grep_a":
        t = 1.0 + 0.1 * torch.randn(shape, generator=g)
    elif leaf == "weight_g":
        t = 0.5 + 0.1 * torch.rand(shape, generator=g)
    elif len(shape) == 1:
        t = 1.0 + 0.1 * torch.randn(shape, generator=g)
    else:
        if scheme == "n002":
            std = 0.02
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            std = 1.0 / math.sqrt(max(fan_in, 1))
            if "embed_tokens" in name or "position_embedding" in name or "query_tokens" in name \
                    or "class_embedding" in name or "relative_attention_bias" in name \
                    or "level_embed" in name or "avs_query" in name or "no_mask_embed" in name:
                std = 0.5
            if "positional_encoding_gaussian_matrix" in name:
                std = 1.0
            # Decoder sub-layer outputs are kept small against the residual stream (as in a trained pre-LN
            # decoder).  A fully random decoder is chaotic: even an exact bf16-storage execution of it moves the
            # logits by ~10 % of their range, which would make greedy-id parity between bf16 and fp32 vacuous.
            if ".o_proj.weight" in name or ".down_proj.weight" in name:
                std *= 0.2
            if ".lora_B" in name:
                std *= 0.1
        t = std * torch.randn(shape, generator=g)
    return t.to(dtype)


def synth_state_dict(shapes: Iterable[Tuple[str, Sequence[int]]], seed: int = 0, scheme: str = "fan_in",
                     dtype: torch.dtype = torch.float32) -> Dict[str, torch.Tensor]:
    return {n: synth_tensor(n, s, seed, scheme, dtype) for n, s in shapes}


def checksum(t: torch.Tensor) -> float:
    """Order-sensitive fp64 checksum used to verify regenerated weights against a fixture."""
    f = t.detach().to(torch.float64).reshape(-1)
    if f.numel() == 0:
        return 0.0
    w = torch.arange(1, f.numel() + 1, dtype=torch.float64).remainder_(977.0).add_(1.0)
    return float((f * w).sum())


# ---------------------------------------------------------------- synthetic clip inputs

def synth_video(t_v: int = 8, seed: int = 42, clip: int = 0, size: int = 224) -> torch.Tensor:
    """[T_v,3,size,size] fp32: uniform u8 pixels -> /255 -> CLIP normalise (the tensor layout
    CLIPImageProcessor yields at reference dataset/quick_start_dataset.py:315-316)."""
    g = _gen(seed, f"video/{clip}")
    px = torch.randint(0, 256, (t_v, 3, size, size), generator=g).float() / 255.0
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    return (px - mean) / std


def synth_audio(t_a: int = 10, l_a: int = 98, seed: int = 42, clip: int = 0, mel: int = 128) -> torch.Tensor:
    """[T_a,L_a,128] fp32 ~ N(0,0.5^2): the statistics of (fbank-15.41663)/(2*6.55582)
    (reference dataset/audio_processor.py:29-41)."""
    g = _gen(seed, f"audio/{clip}")
    return 0.5 * torch.randn((t_a, l_a, mel), generator=g)


def synth_prompt_ids(n_text: int, vocab: int, special: Dict[str, int], seed: int = 42, clip: int = 0) -> torch.Tensor:
    """n_text ids uniform in [3, vocab) with <video_start><video><video_end> and
    <audio_start><audio><audio_end> at fixed offsets.  `special` is SPECIAL_TOKEN_2_IDS."""
    g = _gen(seed, f"ids/{clip}")
    ids = torch.randint(3, vocab, (n_text,), generator=g, dtype=torch.long)
    v0 = max(1, n_text // 8)
    a0 = max(v0 + 4, n_text // 3)
    assert a0 + 3 <= n_text, "prompt too short for both modality blocks"
    ids[v0:v0 + 3] = torch.tensor([special["<video_start>"], special["<video>"], special["<video_end>"]])
    ids[a0:a0 + 3] = torch.tensor([special["<audio_start>"], special["<audio>"], special["<audio_end>"]])
    return ids


def synth_image(h: int, w: int, seed: int) -> np.ndarray:
    """Deterministic uint8 test image with smooth structure + noise (so the bicubic taps matter); regenerated by the tests."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = np.stack([127 + 120 * np.sin(xx / 7.0 + c) * np.cos(yy / (5.0 + c)) for c in range(3)], -1)
    return np.clip(base + rng.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)


def synth_waveform(seconds: float, seed: int, sr: int = 16000) -> np.ndarray:
    """Deterministic float32 waveform in [-1, 1]: a few drifting tones + noise (front-end tests / probes)."""
    rng = np.random.default_rng(seed)
    t = np.arange(int(seconds * sr), dtype=np.float64) / sr
    x = sum(a * np.sin(2 * np.pi * (f + 40 * np.sin(0.7 * t + p)) * t + p)
            for a, f, p in ((0.3, 220.0, 0.1), (0.2, 1333.0, 1.0), (0.1, 4100.0, 2.0)))
    return np.clip(x + rng.normal(0, 0.05, t.shape), -1, 1).astype(np.float32)
