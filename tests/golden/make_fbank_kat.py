#!/usr/bin/env python3
"""Known-answer vectors for the kaldi fbank front-end, computed by a SECOND, independent implementation written from Kaldi's
published definition of `compute-fbank-feats` (feat/feature-window.cc ExtractWindow / ProcessWindow, feat/feature-fbank.cc
FbankComputer::Compute, feat/mel-computations.cc MelBanks) - float64 throughout, scipy.fft.rfft, per-frame loops, no code shared with
oracle/frontend_oracle.py (which restates torchaudio's vectorised float32 port, compliance/kaldi.py) or with the HIP kernel.

torchaudio is not installable here, so the reference's own call (dataset/audio_processor.py:29-41 -> torchaudio.compliance.kaldi.fbank)
cannot be recorded: the fbank stays PARITY UNPINNED in the judge's sense until a torchaudio-generated fixture exists.  What these
vectors do pin is the ALGORITHM: two independently written implementations of Kaldi's definition (this one and the oracle) and the
device kernel must agree on every waveform below, edge cases included.

Options = what the reference passes (num_mel_bins 128, 16 kHz, 25 ms / 10 ms) + Kaldi / torchaudio defaults: dither 0, remove_dc_offset,
preemphasis 0.97, povey window, round_to_power_of_two (512), snip_edges, low_freq 20, high_freq = Nyquist, use_power, use_log_fbank,
use_energy False, energy floor = float epsilon on the mel energies.  The waveform is scaled by 2**15 first (audio_processor.py:33).

    python tests/golden/make_fbank_kat.py        # rewrites tests/golden/fbank_kat.npz
"""
import json
import math
import os
import sys

import numpy as np
from scipy import fft as sfft

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

SR, FRAME_MS, SHIFT_MS, NMEL = 16000, 25.0, 10.0, 128
PREEMPH, LOW_FREQ = 0.97, 20.0


def mel_scale(f):
    return 1127.0 * math.log(1.0 + f / 700.0)


def kaldi_mel_banks(nfft_padded):
    """mel-computations.cc MelBanks::MelBanks (no VTLN): triangular filters equally spaced on the mel axis between low_freq and
    Nyquist; only the first nfft/2 FFT bins take part (Kaldi drops the Nyquist bin)."""
    nbins_fft = nfft_padded // 2
    bin_width = SR / nfft_padded
    mlo, mhi = mel_scale(LOW_FREQ), mel_scale(0.5 * SR)
    delta = (mhi - mlo) / (NMEL + 1)
    W = np.zeros((NMEL, nbins_fft), np.float64)
    for b in range(NMEL):
        left, center, right = mlo + b * delta, mlo + (b + 1) * delta, mlo + (b + 2) * delta
        for i in range(nbins_fft):
            m = mel_scale(bin_width * i)
            if left < m < right:
                W[b, i] = (m - left) / (center - left) if m <= center else (right - m) / (right - center)
    return W


def kaldi_fbank_f64(wave):
    """wave: float64 samples ALREADY scaled to the int16 range.  Returns [frames, 128] log-mel energies (float64)."""
    n = int(SR * FRAME_MS * 0.001)          # 400
    shift = int(SR * SHIFT_MS * 0.001)      # 160
    nfft = 1
    while nfft < n:
        nfft *= 2                           # 512
    if len(wave) < n:
        return np.zeros((0, NMEL))
    nframes = 1 + (len(wave) - n) // shift  # snip_edges
    window = np.array([(0.5 - 0.5 * math.cos(2.0 * math.pi * i / (n - 1))) ** 0.85 for i in range(n)])
    W = kaldi_mel_banks(nfft)
    eps = float(np.finfo(np.float32).eps)
    out = np.zeros((nframes, NMEL))
    for f in range(nframes):
        x = np.array(wave[f * shift: f * shift + n], dtype=np.float64)
        x = x - x.sum() / n                                  # remove_dc_offset
        for i in range(n - 1, 0, -1):                        # feature-window.cc Preemphasize
            x[i] -= PREEMPH * x[i - 1]
        x[0] -= PREEMPH * x[0]
        x = x * window
        spec = sfft.rfft(np.concatenate([x, np.zeros(nfft - n)]))
        power = spec.real ** 2 + spec.imag ** 2              # bins 0 .. nfft/2
        mel = W @ power[: nfft // 2]
        out[f] = np.log(np.maximum(mel, eps))
    return out


def waveforms():
    """name -> float32 waveform in [-1, 1] (what librosa.load hands the reference)."""
    from crab_amd import synth
    rng = np.random.default_rng(20240521)
    t = np.arange(32000) / SR
    w = {
        "tones_2s": synth.synth_waveform(2.0, 5),
        "tones_1s": synth.synth_waveform(1.0, 9),
        "white_noise": rng.uniform(-0.5, 0.5, 32000).astype(np.float32),
        "dc_offset": (0.25 + 0.01 * np.sin(2 * np.pi * 440 * t)).astype(np.float32),
        "silence": np.zeros(16000, np.float32),                                     # every mel energy hits the epsilon floor
        "full_scale_square": np.where(np.sin(2 * np.pi * 1000 * t[:8000]) >= 0, 1.0, -1.0).astype(np.float32),
        "one_frame": synth.synth_waveform(0.025, 3)[:400],                          # exactly 400 samples -> one frame
        "chirp": np.sin(2 * np.pi * (50 + 3900 * t[:24000]) * t[:24000]).astype(np.float32) * np.float32(0.7),
        "impulse": np.concatenate([np.zeros(777, np.float32), np.ones(1, np.float32), np.zeros(2000, np.float32)]),
    }
    return w


def main():
    arrs, meta = {}, {"sr": SR, "scale": 2 ** 15, "names": [], "note": "float64 Kaldi-spec implementation, scipy rfft"}
    for name, w in waveforms().items():
        fb = kaldi_fbank_f64(w.astype(np.float64) * 2 ** 15)
        arrs["wave_" + name] = w
        arrs["fbank_" + name] = fb.astype(np.float32)
        meta["names"].append(name)
        print(f"{name:18s} {len(w):6d} samples -> {fb.shape}  range [{fb.min():.3f}, {fb.max():.3f}]")
    path = os.path.join(HERE, "fbank_kat.npz")
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **arrs)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
