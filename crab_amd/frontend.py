"""Input front-end on the device: mirrors of the two preprocessing calls the reference's dataset code
makes on the CPU right before `generate()`.

  * `CLIPImageProcessor.preprocess(frames, return_tensors='pt')['pixel_values']`  (dataset/quick_start_dataset.py:315,457:
    transformers CLIPImageProcessor of openai/clip-vit-large-patch14: RGB, shortest edge -> 224 with Pillow BICUBIC,
    centre crop 224, 1/255, CLIP mean / std, channels first)
  * `preprocess(source, fbank_mean, fbank_std)`  (dataset/audio_processor.py:29-41: waveform * 2**15 ->
    torchaudio.compliance.kaldi.fbank(num_mel_bins=128, 16 kHz, 25 ms / 10 ms) -> (fbank - mean) / (2 std))

Both run as HIP kernels through the C-ABI (csrc/frontend.hip); decoding media files (decord / librosa) stays outside.
The resize is bit-exact with Pillow (fixed-point taps computed by the library's host helper), the fbank follows the
published torchaudio 2.0.1 algorithm in fp32.  No CPU fallback: without the library the calls raise CrabHipError."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib
from .ops import _dev, _p, _stream

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _as_u8_hwc(img) -> torch.Tensor:
    """PIL.Image / numpy uint8 [H,W,3] / torch uint8 [H,W,3] -> torch uint8 [H,W,3] (device unchanged for tensors)."""
    if isinstance(img, torch.Tensor):
        t = img
    else:
        if hasattr(img, "convert"):                      # PIL: do_convert_rgb
            img = np.asarray(img.convert("RGB"))
        t = torch.from_numpy(np.array(img, copy=True))
    if t.dtype != torch.uint8 or t.dim() != 3 or t.shape[2] != 3:
        raise ValueError("CLIPImageProcessor: expected uint8 RGB images of shape [H, W, 3]")
    return t


class _BatchFeature(dict):
    """`BatchFeature`-like: item and attribute access to 'pixel_values'."""
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class CLIPImageProcessor:
    """Device mirror of transformers' CLIPImageProcessor (image_processing_clip.py, 4.37.2) for uint8 RGB inputs."""

    def __init__(self, size: int = 224, crop_size: int = 224, image_mean=CLIP_MEAN, image_std=CLIP_STD, rescale_factor: float = 1 / 255,
                 device="cuda", dtype=torch.float32):
        if size != crop_size:
            raise NotImplementedError("shortest-edge size and crop size are equal in every CLIP checkpoint Crab uses")
        self.size, self.crop_size = {"shortest_edge": size}, {"height": crop_size, "width": crop_size}
        self.image_mean, self.image_std, self.rescale_factor = tuple(image_mean), tuple(image_std), float(rescale_factor)
        self.device, self.dtype = torch.device(device), dtype
        self._coeffs: Dict[Tuple[int, int], Tuple[torch.Tensor, torch.Tensor, int]] = {}

    @staticmethod
    def resize_size(h: int, w: int, shortest: int) -> Tuple[int, int]:
        """image_transforms.get_resize_output_image_size(size=shortest, default_to_square=False)."""
        short, long = (w, h) if w <= h else (h, w)
        new_short, new_long = shortest, int(shortest * long / short)
        return (new_long, new_short) if w <= h else (new_short, new_long)

    def _coeff(self, in_size: int, out_size: int):
        key = (in_size, out_size)
        if key not in self._coeffs:
            lib = _lib.load()
            ks = lib.crab_bicubic_ksize(in_size, out_size)
            if ks <= 0:
                raise _lib.CrabHipError("crab_bicubic_ksize: bad size")
            bounds = np.zeros((out_size, 2), np.int32)
            kk = np.zeros((out_size, ks), np.int32)
            rc = lib.crab_bicubic_coeffs(in_size, out_size, bounds.ctypes.data_as(C.c_void_p), kk.ctypes.data_as(C.c_void_p), ks)
            if rc != ks:
                raise _lib.CrabHipError(f"crab_bicubic_coeffs failed ({rc})")
            self._coeffs[key] = (torch.from_numpy(bounds).to(self.device), torch.from_numpy(kk).to(self.device), ks)
        return self._coeffs[key]

    def _resample(self, x: torch.Tensor, out_size: int, horizontal: bool) -> torch.Tensor:
        N, H, W, Cc = x.shape
        bounds, kk, ks = self._coeff(W if horizontal else H, out_size)
        out = torch.empty((N, H, out_size, Cc) if horizontal else (N, out_size, W, Cc), device=x.device, dtype=torch.uint8)
        d = _dev(x)
        _lib.check(_lib.load().crab_resample_u8(_lib.ctx(d), _stream(), _p(x), N, H, W, Cc, _p(out), out_size, 1 if horizontal else 0,
                                                _p(bounds), _p(kk), ks), d)
        return out

    def resize_crop(self, x: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
        """uint8 [N,H,W,3] on the device -> (resized uint8 [N,oh,ow,3], crop top, crop left)."""
        s = self.size["shortest_edge"]
        N, H, W, _ = x.shape
        oh, ow = self.resize_size(H, W, s)
        if ow != W:
            x = self._resample(x, ow, True)              # Pillow: horizontal pass first, uint8 in between
        if oh != H:
            x = self._resample(x, oh, False)
        return x, (oh - s) // 2, (ow - s) // 2

    def resize_exact(self, image, height: int, width: int) -> torch.Tensor:
        """`PIL.Image.resize((width, height))` with Pillow's default BICUBIC filter, aspect ignored: the step the reference's image
        tasks apply before the processor (dataset/quick_start_dataset.py:456,488,521).  uint8 [H,W,3] -> uint8 [height,width,3]."""
        x = _as_u8_hwc(image).to(self.device)[None].contiguous()
        if x.shape[2] != width:
            x = self._resample(x, width, True)
        if x.shape[1] != height:
            x = self._resample(x, height, False)
        return x[0]

    def preprocess(self, images, return_tensors: str = "pt", **_unused) -> _BatchFeature:
        """images: one image or a list of PIL images / uint8 [H,W,3] arrays or tensors -> {'pixel_values': [T,3,224,224]}
        on the device.  Images of equal size are processed in one batch of launches."""
        if not isinstance(images, (list, tuple)):
            images = [images]
        ts = [_as_u8_hwc(i) for i in images]
        s = self.crop_size["height"]
        out = torch.empty((len(ts), 3, s, s), device=self.device, dtype=self.dtype)
        groups: Dict[Tuple[int, int], List[int]] = {}
        for i, t in enumerate(ts):
            groups.setdefault((t.shape[0], t.shape[1]), []).append(i)
        mean = (C.c_float * 3)(*self.image_mean)
        std = (C.c_float * 3)(*self.image_std)
        for (_h, _w), idxs in groups.items():
            x = torch.stack([ts[i] for i in idxs], 0).to(self.device).contiguous()
            r, top, left = self.resize_crop(x)
            o = torch.empty((len(idxs), 3, s, s), device=self.device, dtype=self.dtype)
            d = _dev(r)
            _lib.check(_lib.load().crab_clip_normalize(_lib.ctx(d), _stream(), _p(r), r.shape[0], r.shape[1], r.shape[2], top, left, s, _p(o),
                                                       1 if self.dtype == torch.bfloat16 else 0, C.cast(mean, C.c_void_p),
                                                       C.cast(std, C.c_void_p), self.rescale_factor), d)
            out[torch.tensor(idxs, device=self.device)] = o
        return _BatchFeature(pixel_values=out)

    __call__ = preprocess


# ------------------------------------------------------------------------------------------------ audio
def povey_window(n: int = 400) -> np.ndarray:
    """torchaudio kaldi.py _feature_window_function('povey'): hann(n, periodic=False) ** 0.85."""
    i = np.arange(n, dtype=np.float64)
    return ((0.5 - 0.5 * np.cos(2.0 * math.pi * i / (n - 1))) ** 0.85).astype(np.float32)


def mel_banks_t(num_bins: int = 128, padded: int = 512, sample_freq: float = 16000.0, low_freq: float = 20.0) -> np.ndarray:
    """torchaudio kaldi.py get_mel_banks (no VTLN, high_freq = Nyquist) with the zero Nyquist column, TRANSPOSED to
    [padded/2 + 1, num_bins] float32 (the kernel's threads read consecutive mel bins)."""
    num_fft_bins = padded // 2
    high_freq = 0.5 * sample_freq
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)
    mel_low, mel_high = mel(low_freq), mel(high_freq)
    delta = np.float32((mel_high - mel_low) / (num_bins + 1))
    b = np.arange(num_bins, dtype=np.float32)[:, None]
    left, center, right = np.float32(mel_low) + b * delta, np.float32(mel_low) + (b + 1.0) * delta, np.float32(mel_low) + (b + 2.0) * delta
    m = mel(np.float32(sample_freq / padded) * np.arange(num_fft_bins, dtype=np.float32))[None].astype(np.float32)
    bins = np.maximum(np.float32(0.0), np.minimum((m - left) / (center - left), (right - m) / (right - center))).astype(np.float32)
    return np.ascontiguousarray(np.pad(bins, ((0, 0), (0, 1))).T)


_FB_CONST: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}


def _fbank_consts(device: torch.device):
    key = device.index or 0
    if key not in _FB_CONST:
        _FB_CONST[key] = (torch.from_numpy(povey_window()).to(device), torch.from_numpy(mel_banks_t()).to(device))
    return _FB_CONST[key]


def kaldi_fbank(waveform: torch.Tensor, in_scale: float = 1.0, out_sub: float = 0.0, out_scale: float = 1.0,
                preemphasis: float = 0.97) -> torch.Tensor:
    """waveform fp32 [n, L] (device) -> log-mel fbank fp32 [n, 1 + (L-400)//160, 128] with the arguments the reference
    passes to torchaudio.compliance.kaldi.fbank and torchaudio's defaults otherwise."""
    if waveform.dim() == 1:
        waveform = waveform[None]
    w = waveform.to(dtype=torch.float32).contiguous()
    if not w.is_cuda:
        raise _lib.CrabHipError("kaldi_fbank: waveform must be on the GPU")
    n, L = w.shape
    frames = _lib.load().crab_kaldi_fbank_frames(L)
    if frames <= 0:
        raise ValueError("kaldi_fbank: need at least 400 samples (25 ms at 16 kHz)")
    win, mel_t = _fbank_consts(w.device)
    out = torch.empty((n, frames, 128), device=w.device, dtype=torch.float32)
    d = _dev(w)
    _lib.check(_lib.load().crab_kaldi_fbank(_lib.ctx(d), _stream(), _p(w), w.stride(0), n, L, in_scale, preemphasis, _p(win), _p(mel_t),
                                            _p(out), out_sub, out_scale), d)
    return out


def preprocess(source: torch.Tensor, fbank_mean: float = 15.41663, fbank_std: float = 6.55582) -> torch.Tensor:
    """dataset/audio_processor.py:29-41 `preprocess`: source [n, L] waveforms in [-1, 1] -> [n, frames, 128] fp32 on the GPU."""
    src = source if isinstance(source, torch.Tensor) else torch.as_tensor(np.asarray(source))
    return kaldi_fbank(src.to("cuda"), in_scale=float(2 ** 15), out_sub=fbank_mean, out_scale=1.0 / (2 * fbank_std))


def avqa_audio_segments(audio: torch.Tensor, tot: int = 60) -> torch.Tensor:
    """dataset/quick_start_dataset.py:320-336: ten 2 s windows around 0, 6, .., 54 s of a 60 s clip, silence padded -> [10, 2*nps]."""
    length = audio.shape[0]
    nps = int(length / tot)
    segs = []
    for indice in range(0, 60, 6):
        start_time, end_time = max(0, indice - 0.5), min(tot, indice + 1.5)
        seg = audio[int(start_time * nps): int(nps * end_time)]
        pad = 2 * nps - seg.shape[0]
        if indice - 0.5 < 0:
            seg = torch.cat([seg.new_zeros(pad), seg])
        if indice + 1.5 > tot:
            seg = torch.cat([seg, seg.new_zeros(pad)])
        segs.append(seg)
    return torch.stack(segs, 0)
