"""CPU-side checks of the C-ABI boundary: the in-tree library loads and exports every declared symbol.
No compute call is made here (there is no GPU in the build container)."""
import os
import re

import pytest

from crab_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    out = set()
    for f in os.listdir(os.path.join(ROOT, "include")):
        txt = open(os.path.join(ROOT, "include", f)).read()
        out |= set(re.findall(r"^\s*(?:int|int64_t|void|const char\*)\s+(crab_[a-z0-9_]+)\s*\(", txt, flags=re.M))
    return out


def test_library_built_in_tree():
    assert os.path.isfile(_lib.LIB_PATH), "run __graft_entry__.build() first"


def test_every_declared_symbol_is_exported_and_bound():
    lib = _lib.load()
    declared = _declared()
    assert declared, "no declarations found in include/*.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"
    missing = declared - set(_lib.SYMBOLS)
    assert not missing, f"ctypes bindings missing for {missing}"
    assert lib.crab_abi_version() >= 7


def test_ops_fail_loudly_without_gpu_tensors():
    import torch
    from crab_amd import ops
    x = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(_lib.CrabHipError):
        ops.gemm(x, x)


def test_struct_mirrors_match_the_compiled_layout():
    """The ctypes mirrors of crab_gemm_desc / crab_attn_desc must have the size the library was compiled with."""
    import ctypes as C
    lib = _lib.load()
    assert C.sizeof(_lib.GemmDesc) == lib.crab_sizeof_gemm_desc()
    assert C.sizeof(_lib.AttnDesc) == lib.crab_sizeof_attn_desc()
    assert C.sizeof(_lib.LlamaLayer) == lib.crab_sizeof_llama_layer()
    assert C.sizeof(_lib.LlamaIO) == lib.crab_sizeof_llama_io()
    assert C.sizeof(_lib.EncIO) == lib.crab_sizeof_enc_io()
    assert C.sizeof(_lib.ClipLayerW) == lib.crab_sizeof_clip_layer_w()
    assert C.sizeof(_lib.BeatsLayerW) == lib.crab_sizeof_beats_layer_w()
    assert C.sizeof(_lib.QformerLayerW) == lib.crab_sizeof_qformer_layer_w()


def test_entry_points_reject_null_context_and_operands_without_a_gpu():
    """Argument validation happens before any HIP call: exercised here without a device."""
    import ctypes as C
    lib = _lib.load()
    g = _lib.GemmDesc()
    assert lib.crab_gemm_bf16(None, None, C.byref(g)) < 0
    assert lib.crab_rmsnorm(None, None, None, 0, None, None, 0, 1, 8, C.c_float(1e-5)) < 0
    layer, io = _lib.LlamaLayer(), _lib.LlamaIO()
    assert lib.crab_llama_layers(None, None, C.byref(layer), 1, C.byref(io)) < 0
    assert lib.crab_llama_layer_prefill(None, None, C.byref(layer), C.byref(io), 0) < 0
    assert lib.crab_llama_layer_decode(None, None, C.byref(layer), C.byref(io), 0) < 0
    eio = _lib.EncIO()
    assert lib.crab_clip_layer(None, None, C.byref(_lib.ClipLayerW()), C.byref(eio)) < 0
    assert lib.crab_beats_layer(None, None, C.byref(_lib.BeatsLayerW()), C.byref(eio)) < 0
    assert lib.crab_qformer_layer(None, None, C.byref(_lib.QformerLayerW()), C.byref(eio)) < 0
    assert lib.crab_bicubic_ksize(0, 10) < 0 and lib.crab_bicubic_ksize(480, 224) == 2 * 5 + 1
    assert lib.crab_kaldi_fbank_frames(16000) == 98
    assert lib.crab_hyperlora_route_workspace(256, 4096, 48) > 0 and lib.crab_groupnorm_workspace(2, 65536, 32) > 0


def test_prefill_chunk_plan_covers_batch():
    """decoder.GenerationEngine.plan_prefill_chunks: exact partition of the batch, chunks within the row cap, and the
    whole-rounds choice at the AVQA shape (pure host logic; the device is only asked for its CU count)."""
    import types
    from unittest import mock
    import torch
    from crab_amd.decoder import GenerationEngine
    eng = GenerationEngine.__new__(GenerationEngine)
    mk = lambda n, k: types.SimpleNamespace(W=torch.empty((n, k), dtype=torch.bfloat16, device="meta"))
    layer = types.SimpleNamespace(groups=lambda: [mk(12288, 4096), mk(4096, 4096), mk(22016, 4096), mk(4096, 11008)])
    eng.model = types.SimpleNamespace(layers=[layer])
    with mock.patch.object(GenerationEngine, "device", new="cpu", create=True), \
            mock.patch("torch.cuda.get_device_properties", return_value=types.SimpleNamespace(multi_processor_count=256)):
        for B, S in [(1, 5), (3, 100), (16, 702), (256, 702), (255, 559), (7, 4000)]:
            ch = eng.plan_prefill_chunks(B, S)
            assert sum(ch) == B and all(c >= 1 for c in ch)
            assert all(c * S <= 32768 or c == 1 for c in ch)
        assert eng.plan_prefill_chunks(4, 64) == [4]                        # below the ring regime: one chunk
        ch = eng.plan_prefill_chunks(256, 702)
        assert max(ch) > 16 and len(ch) <= 12                                # fuller rounds than the old fixed 16


def test_bench_flop_and_byte_accounting_matches_survey():
    """bench.py's algorithmic work model: SURVEY.md 8d quotes 10.67 TFLOP per AVQA clip for the prefill phase (8 frames, 10 audio
    segments, S = 702) and ~13.7 GB per decode token at batch 1; the config-derived form must reproduce the Llama-2-7B constants."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f = bench.flops_per_clip()
    assert abs(f - 10.67e12) / 10.67e12 < 0.01
    from oracle.crab_oracle import DecoderConfig
    c = DecoderConfig()
    c.hidden_size, c.intermediate_size, c.num_hidden_layers = 4096, 11008, 32
    f2 = bench.flops_per_clip(cfg=c)
    assert abs(f2 - f) / f < 0.01                              # derived linear / LoRA / attention terms == the quoted constants
    b1 = bench.decode_bytes_per_step(1, 830)
    assert 13.5e9 < b1 < 14.2e9
    assert bench.decode_bytes_per_step(256, 830) - bench.decode_bytes_per_step(256, 829) == 256 * 2 * 32 * 4096 * 2
