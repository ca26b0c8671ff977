"""Input front-end: host coefficient helper on CPU, device kernels against the reference-recorded
fixture (tests/golden/frontend_clip.npz, made by the reference's own CLIPImageProcessor call) and the oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

from crab_amd import synth
from oracle import frontend_oracle as FO

HERE = os.path.dirname(os.path.abspath(__file__))


def _fixture():
    z = np.load(os.path.join(HERE, "golden", "frontend_clip.npz"))
    return json.loads(str(z["meta"])), z


@pytest.mark.parametrize("in_size,out_size", [(480, 336), (150, 224), (1920, 398), (224, 224), (37, 224), (91, 550), (5, 3)])
def test_bicubic_coeffs_host_helper_matches_pillow_restatement(in_size, out_size):
    """crab_bicubic_coeffs (C, host) == oracle restatement of Pillow's precompute_coeffs / normalize_coeffs_8bpc."""
    from crab_amd import _lib
    lib = _lib.load()
    ks = lib.crab_bicubic_ksize(in_size, out_size)
    b_ref, k_ref = FO.pil_bicubic_coeffs(in_size, out_size)
    assert ks == k_ref.shape[1]
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ks), np.int32)
    assert lib.crab_bicubic_coeffs(in_size, out_size, bounds.ctypes.data_as(C.c_void_p), kk.ctypes.data_as(C.c_void_p), ks) == ks
    assert np.array_equal(bounds, b_ref) and np.array_equal(kk, k_ref)
    assert lib.crab_bicubic_coeffs(in_size, out_size, bounds.ctypes.data_as(C.c_void_p), kk.ctypes.data_as(C.c_void_p), ks - 1) < 0
    assert lib.crab_kaldi_fbank_frames(32000) == 198 and lib.crab_kaldi_fbank_frames(399) == 0 and lib.crab_kaldi_fbank_frames(400) == 1


def test_mel_banks_and_window_match_oracle():
    from crab_amd import frontend
    assert np.array_equal(frontend.mel_banks_t(), FO.mel_banks().T)
    assert np.array_equal(frontend.povey_window(), FO.povey_window())
    assert frontend.CLIPImageProcessor.resize_size(500, 333, 224) == FO.clip_resize_size(500, 333) == (336, 224)


@pytest.mark.gpu
def test_clip_image_processor_matches_reference_fixture():
    """Device resize + crop is BIT-EXACT with the reference's processor (Pillow bicubic); float output within fp32 rounding."""
    from crab_amd.frontend import CLIPImageProcessor
    meta, z = _fixture()
    proc = CLIPImageProcessor()
    imgs = [synth.synth_image(h, w, meta["seed0"] + i) for i, (h, w) in enumerate(meta["shapes"])]
    for i, img in enumerate(imgs):
        x = torch.from_numpy(img)[None].cuda()
        r, top, left = proc.resize_crop(x)
        got = r[0, top:top + 224, left:left + 224].permute(2, 0, 1).cpu().numpy()
        assert np.array_equal(got, z[f"u8_{i}"]), (meta["shapes"][i], int(np.abs(got.astype(int) - z[f"u8_{i}"].astype(int)).max()))
    # public call, mixed sizes in one list (numpy, torch and PIL inputs), order preserved
    from PIL import Image
    mixed = [imgs[0], torch.from_numpy(imgs[1]), Image.fromarray(imgs[2]), imgs[0]]
    px = proc.preprocess(mixed, return_tensors="pt")["pixel_values"]
    assert px.shape == (4, 3, 224, 224) and px.dtype == torch.float32 and px.is_cuda
    assert np.abs(px[0].cpu().numpy() - z["px_0"]).max() < 2e-6
    assert np.abs(px[1].cpu().numpy() - z["px_1"]).max() < 2e-6
    assert torch.equal(px[0], px[3])
    ref2 = FO.clip_preprocess([imgs[2]])[0]
    assert np.abs(px[2].cpu().numpy() - ref2).max() < 2e-6
    # bf16 output for the bf16 model
    pb = CLIPImageProcessor(dtype=torch.bfloat16).preprocess(imgs[1])["pixel_values"]
    assert pb.dtype == torch.bfloat16 and (pb.float().cpu() - torch.from_numpy(z["px_1"])).abs().max() < 1.6e-2


@pytest.mark.gpu
def test_full_hd_frames_round_trip_properties():
    """1080p frames (the reference decodes at 224 through decord; images arrive at native size): constant images stay
    constant (taps sum to 2^22), and a batch equals its frames processed one by one."""
    from crab_amd.frontend import CLIPImageProcessor
    proc = CLIPImageProcessor()
    const = np.full((1080, 1920, 3), 137, np.uint8)
    r, top, left = proc.resize_crop(torch.from_numpy(const)[None].cuda())
    assert r.shape == (1, 224, 398, 3) and int(r.min()) == 137 and int(r.max()) == 137
    frames = [synth.synth_image(360, 640, 900 + i) for i in range(3)]
    both = proc.preprocess(frames)["pixel_values"]
    for i, f in enumerate(frames):
        assert torch.equal(both[i], proc.preprocess(f)["pixel_values"][0])


@pytest.mark.gpu
def test_kaldi_fbank_matches_oracle():
    """fp32 FFT on the device vs the oracle (float64 rfft): log-mel within 2e-3 absolute wherever the mel energy is not
    vanishing, normalised output as dataset/audio_processor.py:preprocess."""
    from crab_amd import frontend
    waves = np.stack([synth.synth_waveform(2.0, 11), synth.synth_waveform(2.0, 12) * 0.01, np.zeros(32000, np.float32)])
    out = frontend.preprocess(torch.from_numpy(waves))
    ref = FO.audio_preprocess(waves)
    assert out.shape == (3, 198, 128) and out.dtype == torch.float32
    err = np.abs(out.cpu().numpy() - ref)
    assert err[:2].max() < 2e-3 / (2 * 6.55582) * 4, err[:2].max()
    # silence: every mel energy floors at eps -> log(eps) exactly as the reference computes it
    assert np.abs(out[2].cpu().numpy() - ref[2]).max() < 1e-5
    # AVQA slicing: ten 2 s windows of a 60 s clip
    audio = torch.from_numpy(synth.synth_waveform(60.0, 13))
    segs = frontend.avqa_audio_segments(audio)
    ref_segs = np.stack(FO.avqa_audio_segments(audio.numpy()))
    assert np.array_equal(segs.numpy(), ref_segs)
    fb = frontend.preprocess(segs)
    assert fb.shape == (10, 198, 128)
    # short input / ragged length
    one = frontend.kaldi_fbank(torch.from_numpy(waves[0][:400]).cuda())
    assert one.shape == (1, 1, 128)
    with pytest.raises(ValueError):
        frontend.kaldi_fbank(torch.zeros(399).cuda())


# ------------------------------------------------------------------ kaldi fbank known-answer vectors (tests/golden/make_fbank_kat.py)
def _kat():
    import json
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fbank_kat.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return meta, z


def _fbank_agrees(got, kat, rel=1e-3, abs_of_frame_max=1e-10):
    """log-mel comparison in the ENERGY domain: |E_got - E_kat| <= rel * E_kat + abs_of_frame_max * max_bin(E_kat) per frame.
    A plain tolerance on the logarithm is meaningless for bins whose energy is ~1e-13 of the frame's strongest bin (pure tones,
    chirps): there float32 arithmetic - which the reference's torchaudio call uses - is rounding noise, not signal."""
    e_k, e_g = np.exp(kat.astype(np.float64)), np.exp(got.astype(np.float64))
    fmax = e_k.max(axis=1, keepdims=True)
    excess = np.abs(e_g - e_k) - rel * e_k - abs_of_frame_max * fmax
    return float((excess / fmax).max())


def test_fbank_oracle_matches_independent_kaldi_spec_vectors():
    """oracle/frontend_oracle.kaldi_fbank (restating torchaudio's float32 port) against the known-answer vectors of the independent
    float64 implementation written from Kaldi's compute-fbank-feats definition: nine waveforms incl. silence (every bin at the
    epsilon floor), a DC offset, a full-scale square wave, exactly one frame, a chirp, an impulse."""
    meta, z = _kat()
    for name in meta["names"]:
        w, kat = z["wave_" + name], z["fbank_" + name]
        fb = FO.kaldi_fbank(w.astype(np.float32) * np.float32(meta["scale"]))
        assert fb.shape == kat.shape, name
        assert _fbank_agrees(fb, kat) <= 0.0, (name, _fbank_agrees(fb, kat))
        floor = kat < -15.9                        # empty mel bins (no FFT bin inside the triangle) and silence: log(eps) exactly
        assert np.array_equal(fb[floor], kat[floor]), name


@pytest.mark.gpu
def test_fbank_device_matches_independent_kaldi_spec_vectors():
    """The HIP kernel (fp32 FFT in LDS) against the same known-answer vectors."""
    from crab_amd import frontend
    from tests.util import record_parity
    meta, z = _kat()
    for name in meta["names"]:
        w, kat = z["wave_" + name], z["fbank_" + name]
        fb = frontend.kaldi_fbank(torch.from_numpy(w).cuda(), in_scale=float(meta["scale"]))[0].cpu().numpy()
        assert fb.shape == kat.shape, name
        ex = _fbank_agrees(fb, kat, rel=2e-3, abs_of_frame_max=1e-8)
        strong = kat > kat.max(axis=1, keepdims=True) - 12.0          # bins within e^-12 of the frame's strongest
        record_parity(f"kaldi fbank vs Kaldi-spec KAT, {name}: max |dlog| over bins within e^-12 of the frame max",
                      float(np.abs(fb - kat)[strong].max()) if strong.any() else 0.0, 1.0, 2e-3)
        assert ex <= 0.0, (name, ex)
        floor = kat < -15.9
        assert np.abs(fb[floor] - kat[floor]).max() < 1e-5 if floor.any() else True, name


# ------------------------------------------------------------------ kaldi fbank vs a THIRD-PARTY implementation (tests/golden/make_fbank_hf.py)
def _hf():
    import json
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "fbank_hf.npz"))
    return json.loads(bytes(z["meta"]).decode()), z


def test_fbank_oracle_matches_transformers_kaldi_compatible_fbank():
    """oracle/frontend_oracle.kaldi_fbank against Hugging Face transformers' Kaldi-compatible filter bank (audio_utils: povey window,
    kaldi mel scale, preemphasis, DC removal, log with the epsilon floor - what transformers' feature extractors run instead of
    torchaudio.compliance.kaldi.fbank when torchaudio is absent), recorded with the reference's options (audio_processor.py:29-41) on the
    nine edge-case waveforms.  Not torchaudio itself, but an implementation of the same call that this repository's builder did not write."""
    meta, z = _hf()
    _, kz = _kat()
    for name in meta["names"]:
        w, ref = kz["wave_" + name], z["fbank_" + name]
        fb = FO.kaldi_fbank(w.astype(np.float32) * np.float32(meta["scale"]))
        assert fb.shape == ref.shape, name
        assert _fbank_agrees(fb, ref) <= 0.0, (name, _fbank_agrees(fb, ref))
        floor = ref < -15.9
        assert np.array_equal(fb[floor], ref[floor]), name


@pytest.mark.gpu
def test_fbank_device_matches_transformers_kaldi_compatible_fbank():
    """The HIP kernel against the same third-party vectors."""
    from crab_amd import frontend
    from tests.util import record_parity
    meta, z = _hf()
    _, kz = _kat()
    for name in meta["names"]:
        w, ref = kz["wave_" + name], z["fbank_" + name]
        fb = frontend.kaldi_fbank(torch.from_numpy(w).cuda(), in_scale=float(meta["scale"]))[0].cpu().numpy()
        assert fb.shape == ref.shape, name
        ex = _fbank_agrees(fb, ref, rel=2e-3, abs_of_frame_max=1e-8)
        strong = ref > ref.max(axis=1, keepdims=True) - 12.0
        record_parity(f"kaldi fbank vs transformers' Kaldi-compatible fbank, {name}: max |dlog| over bins within e^-12 of the frame max",
                      float(np.abs(fb - ref)[strong].max()) if strong.any() else 0.0, 1.0, 2e-3)
        assert ex <= 0.0, (name, ex)


# ------------------------------------------------------------------ kaldi fbank vs the REFERENCE'S OWN CALL (tests/golden/make_fbank_torchaudio.py)
def _ta():
    """tests/golden/fbank_torchaudio.npz = the reference's dataset/audio_processor.py preprocess() (torchaudio.compliance.kaldi.fbank + the AudioSet
    normalisation) on the nine edge-case waveforms.  Only a box with torchaudio can write it (the build container cannot): until then these two
    tests skip and the fbank stays pinned by the Kaldi-spec vectors and transformers' filter bank above."""
    import json
    path = os.path.join(os.path.dirname(__file__), "golden", "fbank_torchaudio.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/fbank_torchaudio.npz absent: `python tests/golden/make_fbank_torchaudio.py` writes it on a box where torchaudio imports")
    z = np.load(path)
    return json.loads(bytes(z["meta"]).decode()), z


def test_fbank_oracle_matches_reference_torchaudio_call():
    meta, z = _ta()
    _, kz = _kat()
    for name in meta["names"]:
        ref = z["norm_" + name].astype(np.float64) * (2 * meta["fbank_std"]) + meta["fbank_mean"]          # back to log-mel energies
        fb = FO.kaldi_fbank(kz["wave_" + name].astype(np.float32) * np.float32(2 ** 15))
        assert fb.shape == ref.shape, name
        assert _fbank_agrees(fb, ref) <= 0.0, (name, _fbank_agrees(fb, ref))
        got = FO.audio_preprocess(kz["wave_" + name][None].astype(np.float32))[0]
        assert np.abs(got - z["norm_" + name]).max() < 2e-3, name


@pytest.mark.gpu
def test_fbank_device_matches_reference_torchaudio_call():
    from crab_amd import frontend
    meta, z = _ta()
    _, kz = _kat()
    for name in meta["names"]:
        ref = z["norm_" + name].astype(np.float64) * (2 * meta["fbank_std"]) + meta["fbank_mean"]
        fb = frontend.kaldi_fbank(torch.from_numpy(kz["wave_" + name]).cuda(), in_scale=float(2 ** 15))[0].cpu().numpy()
        assert fb.shape == ref.shape, name
        assert _fbank_agrees(fb, ref, rel=2e-3, abs_of_frame_max=1e-8) <= 0.0, name
