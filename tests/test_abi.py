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


def _prototypes():
    """name -> (return type, [parameter types]) of every function declared in include/*.h, each reduced to its ctypes kind"""
    def kind(t):
        t = t.strip()
        if "*" in t or t.endswith("]"):
            return "ptr"
        t = re.sub(r"\b(const|unsigned|signed)\b", "", t).split()
        base = t[0] if t else "int"                                  # "unsigned" alone
        return {"int": "i32", "int32_t": "i32", "int64_t": "i64", "uint64_t": "i64", "long": "i64", "float": "f32", "double": "f64", "void": "void",
                "char": "ptr", "uint32_t": "i32", "size_t": "i64"}[base]
    out = {}
    for f in os.listdir(os.path.join(ROOT, "include")):
        txt = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", f)).read(), flags=re.S)
        txt = re.sub(r"//[^\n]*", " ", txt)
        for m in re.finditer(r"(?:^|[;}\n])\s*((?:const\s+)?(?:int|int64_t|void|char\s*\*|float\s*\*?)\s*\*?)\s*(crab_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
            ret, name, params = m.group(1), m.group(2), m.group(3).strip()
            ps = [] if params in ("", "void") else [re.sub(r"\s+[A-Za-z_][A-Za-z0-9_]*\s*(\[\s*\])?\s*$", lambda mm: " *" if mm.group(1) else "", p.strip()) if not p.strip().endswith("*") else p
                                                    for p in params.split(",")]
            out[name] = (kind(ret), [kind(p) for p in ps])
    return out


def test_ctypes_signatures_match_the_header_prototypes():
    """r06: a binding whose argtypes drift from the prototype (an int for an int64_t stride, a float for a double) does not fail - it passes garbage.
    Every entry of crab_amd/_lib.py SYMBOLS is compared, parameter by parameter, with the declaration in include/crab_hip.h."""
    import ctypes as C
    protos = _prototypes()
    k = {C.c_void_p: "ptr", C.c_char_p: "ptr", C.c_int: "i32", C.c_int32: "i32", C.c_uint32: "i32", C.c_int64: "i64", C.c_uint64: "i64", C.c_long: "i64",
         C.c_float: "f32", C.c_double: "f64", None: "void"}
    checked = 0
    for name, (res, args) in _lib.SYMBOLS.items():
        assert name in protos, f"{name} bound in _lib.py but not declared in include/"
        pres, pargs = protos[name]
        got = [k.get(a, "ptr") for a in args]                        # POINTER(struct) and friends are pointers
        assert got == pargs, f"{name}: ctypes {got} vs header {pargs}"
        assert k.get(res, "ptr") == pres or (pres == "ptr" and k.get(res, "ptr") == "ptr"), f"{name}: restype {res} vs header {pres}"
        checked += 1
    assert checked >= 100
