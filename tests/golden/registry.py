"""Which generator writes which fixture (pure data: imported by make_golden.py and by tests/test_oracle_golden.py, which fails when a
committed .npz has no generator here - a fixture nobody can regenerate in one command rots).

`python tests/golden/make_golden.py` with no argument runs EVERY entry, each in its own interpreter (the reference's modules and the
transformers-5.15 hidden-state recorder carry process-wide state: an encoder pass after any generate() returns other tensors, the harness
generator stubs decord / librosa / cv2 - so generators are only order-independent when they do not share a process).

kind "ref":    a golden_* function of make_golden.py that imports /root/reference;
kind "script": a stand-alone script in this directory (its own docstring says what it pins and why it is separate);
kind "script_optional": the same, but its fixture may be absent (the script exits 3 where it cannot run)."""

GENERATORS = {
    # name: (kind, entry, [fixtures it writes])
    "lora": ("ref", "build_lora_linear", ["hyperlora_linear"]),
    "beats": ("ref", "golden_beats", ["beats_tiny", "beats_buckets"]),
    "clip": ("ref", "golden_clip", ["clip_tiny"]),
    "proj": ("ref", "golden_projectors", ["projectors_tiny"]),
    "full": ("ref", "golden_full", ["full_tiny_llama", "forward_masked_tiny_llama"]),
    "qwen": ("ref", "golden_qwen", ["decoder_tiny_qwen2"]),
    "seg": ("ref", "golden_seg", ["seg_tiny"]),
    "frontend": ("ref", "golden_frontend", ["frontend_clip"]),
    "vqgan": ("ref", "golden_vqgan", ["vqgan_tiny"]),
    "harness": ("ref", "golden_harness", ["harness"]),
    "llama_ops": ("ref", "golden_llama_ops", ["llama_ops"]),
    "qwen_ops": ("ref", "golden_qwen_ops", ["qwen_ops"]),
    "full_qwen": ("ref", "golden_full_qwen", ["full_tiny_qwen"]),
    "id_stats": ("ref", "golden_id_stats", ["id_stats_tiny_llama"]),
    "holes": ("ref", "golden_holes", ["forward_holes_tiny_llama"]),
    "sharp": ("ref", "golden_sharp", ["sharp_tiny_llama"]),
    "avs_loop": ("ref", "golden_avs_loop", ["avs_loop_tiny"]),
    "metrics": ("ref", "golden_metrics", ["seg_metrics"]),
    # full-width, shallow (r06): the benchmarked kernel instantiations pinned to the reference
    "llama_layer_wide": ("ref", "golden_llama_layer_wide", ["llama_layer_wide"]),
    "qwen_layer_wide": ("ref", "golden_qwen_layer_wide", ["qwen_layer_wide"]),
    "llama_decode_regimes_wide": ("ref", "golden_llama_decode_regimes_wide", ["llama_decode_regimes_wide"]),
    "clip_wide": ("ref", "golden_clip_wide", ["clip_wide"]),
    "beats_wide": ("ref", "golden_beats_wide", ["beats_wide"]),
    "projectors_wide": ("ref", "golden_projectors_wide", ["projectors_wide"]),
    "seg_wide": ("ref", "golden_seg_wide", ["seg_wide"]),
    # stand-alone scripts
    "ckpt_manifest": ("script", "make_ckpt_manifest.py", ["ckpt_manifest"]),
    "fbank_hf": ("script", "make_fbank_hf.py", ["fbank_hf"]),
    "fbank_kat": ("script", "make_fbank_kat.py", ["fbank_kat"]),
    # needs torchaudio (absent in the build container): writes its fixture only where `import torchaudio` works, exit code 3 otherwise
    "fbank_torchaudio": ("script_optional", "make_fbank_torchaudio.py", ["fbank_torchaudio"]),
}

# `make_golden.py fullwidth` = the six full-width generators
GROUPS = {"fullwidth": ["llama_layer_wide", "qwen_layer_wide", "llama_decode_regimes_wide", "clip_wide", "beats_wide", "projectors_wide", "seg_wide"]}


def fixtures(optional: bool = False):
    """Fixtures every checkout must hold (optional=True: also those that only some boxes can generate)."""
    return sorted(f for kind, _, fs in GENERATORS.values() for f in fs if optional or kind != "script_optional")
