#!/usr/bin/env python3
"""The kaldi fbank pinned to the REFERENCE'S OWN CALL - for the first box that has torchaudio (SURVEY.md 8 f-3's open caveat).

torchaudio cannot be installed in the build container (no network), so the fbank front-end is pinned only by two substitutes: an independent
Kaldi-spec implementation (make_fbank_kat.py) and transformers' Kaldi-compatible filter bank (make_fbank_hf.py).  This script closes the gap
without new code the day `import torchaudio` works: it imports the reference's dataset/audio_processor.py from /root/reference (librosa stubbed:
only `preprocess` is used) and records what `preprocess(waveform)` - i.e. torchaudio.compliance.kaldi.fbank(waveform * 2**15, num_mel_bins=128,
sample_frequency=16000, frame_length=25, frame_shift=10) followed by (x - 15.41663) / (2 * 6.55582), audio_processor.py:29-41 - returns for the
nine edge-case waveforms of fbank_kat.npz, into tests/golden/fbank_torchaudio.npz.  tests/test_frontend.py prefers that file when it exists
(test_fbank_*_reference_torchaudio_call) and says so when it does not.

    python tests/golden/make_fbank_torchaudio.py       # exit code 3 (nothing written) when torchaudio is not importable
"""
import importlib.machinery
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("CRAB_REFERENCE", "/root/reference")


def main():
    try:
        import torchaudio
        import torchaudio.compliance.kaldi  # noqa: F401
    except Exception as e:      # noqa: BLE001
        print(f"torchaudio is not importable here ({type(e).__name__}: {e}): tests/golden/fbank_torchaudio.npz NOT written; the fbank stays pinned by "
              "fbank_kat.npz / fbank_hf.npz only")
        return 3
    import torch
    if "librosa" not in sys.modules:                     # audio_processor.py imports it at module level; preprocess() never calls it
        m = types.ModuleType("librosa")
        m.__spec__ = importlib.machinery.ModuleSpec("librosa", None)
        sys.modules["librosa"] = m
    sys.path.insert(0, REF)
    from dataset.audio_processor import preprocess
    z = np.load(os.path.join(HERE, "fbank_kat.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    out = {}
    for name in meta["names"]:
        w = torch.from_numpy(z["wave_" + name].astype(np.float32))[None]          # [1, L] in [-1, 1]: preprocess scales by 2**15 itself
        out["norm_" + name] = preprocess(w)[0].numpy().astype(np.float32)          # (fbank - 15.41663) / (2 * 6.55582)
    m = dict(names=meta["names"], torchaudio=torchaudio.__version__, torch=torch.__version__, fbank_mean=15.41663, fbank_std=6.55582,
             note="the reference's dataset/audio_processor.py preprocess() on the waveforms of fbank_kat.npz")
    np.savez_compressed(os.path.join(HERE, "fbank_torchaudio.npz"), meta=np.frombuffer(json.dumps(m).encode(), dtype=np.uint8), **out)
    print("wrote", os.path.join(HERE, "fbank_torchaudio.npz"))
    return 0


if __name__ == "__main__":
    sys.exit(main())
