"""CPU restatement of Crab's input front-end -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module; the product path
(crab_amd/frontend.py -> libcrab_hip.so) never does.

What the reference does (dataset/quick_start_dataset.py:299-343, dataset/audio_processor.py:29-41):
  * video / image: decord / PIL decode (out of scope) -> `CLIPImageProcessor.preprocess(frames)` (transformers 4.37.2,
    openai/clip-vit-large-patch14 preprocessor config): convert RGB, resize shortest edge to 224 with PIL BICUBIC,
    centre crop 224x224, rescale 1/255, normalise by the CLIP mean / std, channels first.
  * audio: librosa load at 16 kHz (out of scope) -> ten 2 s segments -> `preprocess`: waveform * 2**15 ->
    `torchaudio.compliance.kaldi.fbank(num_mel_bins=128, sample_frequency=16000, frame_length=25, frame_shift=10)` ->
    (fbank - 15.41663) / (2 * 6.55582).

Third-party algorithms restated here (absent from /root/reference):
  * Pillow 10.4.0 `src/libImaging/Resample.c` (precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc /
    Vertical_8bpc, bicubic_filter a = -0.5).  PINNED: Pillow is importable in the build container, tests/golden/
    frontend_clip.npz holds outputs of the reference's own call (CLIPImageProcessor.preprocess) and this restatement is
    bit-exact against them (tests/test_oracle_golden.py).
  * torchaudio 2.0.1 `compliance/kaldi.py` (fbank, get_mel_banks, _get_window, povey window).  PARITY UNPINNED:
    torchaudio is not installed in the build container, so no output of the reference's own call could be recorded.  The
    restatement follows the published source; it is checked against known-answer vectors (tests/golden/fbank_kat.npz) produced by
    a second implementation written independently from Kaldi's compute-fbank-feats definition in float64 (tests/golden/
    make_fbank_kat.py: per-frame loops, scipy rfft, nine waveforms incl. silence / DC / square wave / one frame / chirp / impulse),
    which pins the algorithm but not torchaudio's rounding: it stays "unpinned" until a torchaudio-generated fixture exists.
    r03: additionally checked against a THIRD-PARTY implementation of the same call that runs here - Hugging Face transformers'
    Kaldi-compatible filter bank (transformers.audio_utils: what its feature extractors use instead of torchaudio.compliance.kaldi
    when torchaudio is absent), recorded with the reference's options on the same waveforms (tests/golden/make_fbank_hf.py ->
    fbank_hf.npz).  That is not torchaudio's own output either, so the status word stays "parity unpinned"; the evidence is now two
    independent implementations (one not the builder's) agreeing with this restatement and with the device kernel.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2            # Resample.c: coefficients are 22-bit fixed point for 8-bit channels


# ------------------------------------------------------------------------------------------------ Pillow bicubic
def _bicubic(x: float, a: float = -0.5) -> float:
    """Resample.c bicubic_filter."""
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pil_bicubic_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the whole-image box: (bounds [out,2] int32 = first tap,
    tap count; kk [out, ksize] int32 fixed-point taps)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        ww = 0.0
        for x in range(xmax):
            w = _bicubic((x + xmin - center + 0.5) * ss)
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    ki = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << PRECISION_BITS)), np.trunc(0.5 + kk * (1 << PRECISION_BITS)))
    return bounds, ki.astype(np.int32)


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    bounds, ki = pil_bicubic_coeffs(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.zeros((out_size,) + src.shape[1:], np.int64)
    for xx in range(out_size):
        x0, n = bounds[xx]
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(ki[xx, :n].astype(np.int64), src[x0:x0 + n], axes=(0, 0))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255)
    return np.moveaxis(out.astype(np.uint8), 0, axis)


def pil_resize_bicubic(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """`Image.resize((out_w, out_h), BICUBIC)` on a uint8 [H,W,C] array: horizontal pass, then vertical, uint8 between."""
    out = img
    if out_w != img.shape[1]:
        out = _resample_axis(out, out_w, 1)
    if out_h != img.shape[0]:
        out = _resample_axis(out, out_h, 0)
    return out


def clip_resize_size(h: int, w: int, shortest: int = 224) -> Tuple[int, int]:
    """transformers image_transforms.get_resize_output_image_size(size=224, default_to_square=False)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = shortest, int(shortest * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def clip_resize_crop(img: np.ndarray, size: int = 224) -> np.ndarray:
    """uint8 [H,W,3] -> uint8 [224,224,3]: shortest-edge bicubic resize + centre crop (image_processing_clip.py)."""
    oh, ow = clip_resize_size(img.shape[0], img.shape[1], size)
    r = pil_resize_bicubic(img, oh, ow)
    top, left = (oh - size) // 2, (ow - size) // 2
    return r[top:top + size, left:left + size]


def clip_preprocess(images: Sequence[np.ndarray], size: int = 224) -> np.ndarray:
    """The reference's `video_processor.preprocess(frames, return_tensors='pt')['pixel_values']`: float32 [T,3,224,224]."""
    out = []
    mean = np.asarray(CLIP_MEAN, np.float32)
    std = np.asarray(CLIP_STD, np.float32)
    for img in images:
        c = clip_resize_crop(np.asarray(img, np.uint8), size).astype(np.float32) * np.float32(1.0 / 255.0)
        out.append(((c - mean) / std).transpose(2, 0, 1))
    return np.stack(out, 0)


# ------------------------------------------------------------------------------------------------ kaldi fbank
def povey_window(n: int = 400) -> np.ndarray:
    """kaldi.py _feature_window_function('povey'): hann(n, periodic=False) ** 0.85 (float32)."""
    i = np.arange(n, dtype=np.float64)
    return ((0.5 - 0.5 * np.cos(2.0 * math.pi * i / (n - 1))) ** 0.85).astype(np.float32)


def mel_banks(num_bins: int = 128, padded: int = 512, sample_freq: float = 16000.0, low_freq: float = 20.0,
              high_freq: float = 0.0) -> np.ndarray:
    """kaldi.py get_mel_banks (no VTLN) + the zero Nyquist column fbank() pads on: float32 [num_bins, padded/2 + 1]."""
    num_fft_bins = padded // 2
    nyquist = 0.5 * sample_freq
    if high_freq <= 0.0:
        high_freq += nyquist
    fft_bin_width = sample_freq / padded
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = np.arange(num_bins, dtype=np.float32)[:, None]
    left = np.float32(mel_low) + b * np.float32(delta)
    center = np.float32(mel_low) + (b + 1.0) * np.float32(delta)
    right = np.float32(mel_low) + (b + 2.0) * np.float32(delta)
    m = mel(np.float32(fft_bin_width) * np.arange(num_fft_bins, dtype=np.float32))[None].astype(np.float32)
    up = (m - left) / (center - left)
    down = (right - m) / (right - center)
    bins = np.maximum(np.float32(0.0), np.minimum(up, down)).astype(np.float32)
    return np.pad(bins, ((0, 0), (0, 1)))


def kaldi_fbank(waveform: np.ndarray, num_mel_bins: int = 128, sample_frequency: float = 16000.0, frame_length: float = 25.0,
                frame_shift: float = 10.0, preemphasis: float = 0.97, dtype=np.float32) -> np.ndarray:
    """kaldi.py fbank with the reference's arguments and torchaudio's defaults (dither 0, remove_dc_offset, povey window,
    round_to_power_of_two, snip_edges, use_power, use_log_fbank, energy not used): waveform [L] -> [m, num_mel_bins]."""
    x = np.asarray(waveform, dtype)
    win = int(sample_frequency * frame_length * 0.001)          # 400
    shift = int(sample_frequency * frame_shift * 0.001)         # 160
    padded = 1 << (win - 1).bit_length()                        # 512
    if x.shape[0] < win:
        return np.zeros((0, num_mel_bins), dtype)
    m = 1 + (x.shape[0] - win) // shift
    idx = np.arange(m)[:, None] * shift + np.arange(win)[None]
    fr = x[idx]
    fr = fr - fr.mean(axis=1, keepdims=True, dtype=dtype)
    prev = np.concatenate([fr[:, :1], fr[:, :-1]], 1)            # replicate-pad left by one sample
    fr = fr - dtype(preemphasis) * prev
    fr = fr * povey_window(win).astype(dtype)
    fr = np.pad(fr, ((0, 0), (0, padded - win)))
    spec = np.fft.rfft(fr.astype(np.float64), axis=1)
    power = (spec.real ** 2 + spec.imag ** 2).astype(dtype)
    mel = power @ mel_banks(num_mel_bins, padded, sample_frequency).astype(dtype).T
    eps = np.finfo(np.float32).eps
    return np.log(np.maximum(mel, dtype(eps))).astype(dtype)


def audio_preprocess(source: np.ndarray, fbank_mean: float = 15.41663, fbank_std: float = 6.55582) -> np.ndarray:
    """dataset/audio_processor.py:29-41 preprocess: source [n, L] float waveforms in [-1,1] -> [n, m, 128] float32."""
    out = [kaldi_fbank(np.asarray(w, np.float32) * np.float32(2 ** 15)) for w in source]
    fb = np.stack(out, 0)
    return ((fb - np.float32(fbank_mean)) / np.float32(2 * fbank_std)).astype(np.float32)


def avqa_audio_segments(audio: np.ndarray, tot: int = 60) -> List[np.ndarray]:
    """quick_start_dataset.py:320-336: ten 2 s windows [i-0.5, i+1.5) s around i = 0, 6, .., 54 of a 60 s clip, silence padded."""
    length = len(audio)
    nps = int(length / tot)
    segs = []
    for indice in range(0, 60, 6):
        start_time, end_time = max(0, indice - 0.5), min(tot, indice + 1.5)
        seg = audio[int(start_time * nps): int(nps * end_time)]
        if indice - 0.5 < 0:
            seg = np.concatenate((np.zeros(2 * nps - len(seg), dtype=seg.dtype), seg), 0)
        if indice + 1.5 > tot:
            seg = np.concatenate((seg, np.zeros(2 * nps - len(seg), dtype=seg.dtype)), 0)
        segs.append(seg)
    return segs
