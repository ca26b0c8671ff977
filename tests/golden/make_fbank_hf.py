#!/usr/bin/env python3
"""Third-party pin for the kaldi fbank front-end: Hugging Face transformers' own Kaldi-compatible filter bank (transformers.audio_utils:
window_function("povey"), mel_filter_bank(mel_scale="kaldi", triangularize_in_mel_space=True), spectrogram(preemphasis, remove_dc_offset,
log_mel="log", mel_floor = float epsilon)) - the recipe transformers' feature extractors use IN PLACE OF torchaudio.compliance.kaldi.fbank
when torchaudio is not installed (e.g. SeamlessM4TFeatureExtractor: "waveform * 2**15  # Kaldi compliance: 16-bit signed integers"), called
here with the options the reference passes (dataset/audio_processor.py:29-41: num_mel_bins 128, 16 kHz, 25 ms / 10 ms, everything else
torchaudio's defaults: dither 0, povey window, preemphasis 0.97, remove_dc_offset, 512-point FFT, low_freq 20, high_freq = Nyquist,
snip_edges).  torchaudio itself cannot be installed here (no network), so this is NOT the reference's own call - but it is an
implementation of that call written and maintained by a third party, not by this repository's builder; the waveforms are the ones of
tests/golden/make_fbank_kat.py (builder-written Kaldi-spec vectors) so both pins cover the same edge cases.

    python tests/golden/make_fbank_hf.py        # rewrites tests/golden/fbank_hf.npz
"""
import json
import os

import numpy as np
import transformers
from transformers.audio_utils import mel_filter_bank, spectrogram, window_function

HERE = os.path.dirname(os.path.abspath(__file__))


def hf_kaldi_fbank(wave_int16_scale: np.ndarray) -> np.ndarray:
    window = window_function(400, "povey", periodic=False)
    mel = mel_filter_bank(num_frequency_bins=257, num_mel_filters=128, min_frequency=20, max_frequency=8000, sampling_rate=16000, norm=None,
                          mel_scale="kaldi", triangularize_in_mel_space=True)
    return spectrogram(wave_int16_scale, window, frame_length=400, hop_length=160, fft_length=512, power=2.0, center=False, preemphasis=0.97,
                       mel_filters=mel, log_mel="log", mel_floor=1.192092955078125e-07, remove_dc_offset=True).T


def main():
    z = np.load(os.path.join(HERE, "fbank_kat.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    out = {}
    for name in meta["names"]:
        w = z["wave_" + name].astype(np.float64) * meta["scale"]
        fb = hf_kaldi_fbank(w)
        out["fbank_" + name] = fb.astype(np.float32)
        kat = z["fbank_" + name]
        strong = kat > kat.max(axis=1, keepdims=True) - 12.0
        print(f"{name:20s} {fb.shape}  max |hf - builder KAT| over strong bins {np.abs(fb - kat)[strong].max():.3e}")
    m = dict(names=meta["names"], scale=meta["scale"], transformers=transformers.__version__,
             note="transformers.audio_utils Kaldi-compatible fbank (povey window, kaldi mel scale, preemphasis 0.97, remove_dc_offset, log, eps floor)")
    np.savez_compressed(os.path.join(HERE, "fbank_hf.npz"), meta=np.frombuffer(json.dumps(m).encode(), dtype=np.uint8), **out)
    print("wrote", os.path.join(HERE, "fbank_hf.npz"))


if __name__ == "__main__":
    main()
