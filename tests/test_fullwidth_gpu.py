"""The benchmarked kernel instantiations against REFERENCE-generated numbers (tests/golden/*_wide.npz, `make_golden.py fullwidth`):
one hyper-LoRA Llama-2-7B-wide layer and one Qwen2-7B-wide layer (prefill through the 256 x 256 ring GEMMs incl. K = 11008 / 18944, the
fused d = 128 RoPE epilogue and the 128-row flash forward; cached decode steps through attn_decode_kernel<128> / attn_decode_gqa_kernel<128, 7>
and the split-K RoPE epilogue), a CLIP-L/14-wide tower on 36 frames, BEATs at 768 / 12 heads with the gated bias at n = 48 and 96, both Q-Former
projectors at their real configuration, the SegModule under d_model 4096.

Every comparison is against the FIXTURE (fp32 reference output), and its bound is COMPUTED here, not a constant: the same layer through the
oracle's bf16-operand floor (what no bf16-MFMA implementation can beat) and through its bf16-storage emulation (the HIP path's storage points,
exact arithmetic between them), both measured against the same fixture - the HIP error may not exceed 1.5 x the larger of the two.
The library's launch trace (crab_trace_begin / _end) proves WHICH kernels the comparison went through."""
import pytest
import torch

from tests.util import load_fixture, weights_from_table

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
FACTOR = 1.5


def _odev():
    import os
    torch.backends.cuda.matmul.allow_tf32 = False
    return torch.device(os.environ.get("CRAB_ORACLE_DEVICE", "cuda"))


def _err(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert torch.isfinite(got).all()
    return (got - ref).abs().max().item()


def _bounded(what, hip, floor, emu, scale, kernels=None, **extra):
    """hip / floor / emu: max abs distances from the fixture.  Records the triplet and asserts the computed bound."""
    from tests.util import record_parity
    bound = FACTOR * max(floor, emu)
    record_parity(what, hip, scale, bound / scale, bf16_operand_floor_abs=floor, bf16_storage_emulation_abs=emu, hip_over_floor=hip / max(floor, 1e-30),
                  kernels=kernels, **extra)
    assert hip <= bound, (what, f"HIP {hip:.3e} > {FACTOR} x max(floor {floor:.3e}, storage emulation {emu:.3e})", scale)


def _wide_layer(fixture, qwen, prefill_kernels, decode_kernels):
    from crab_amd import ops
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from oracle import crab_oracle as O
    from tests.util import stored_params, wide_layer_inputs
    meta, A = load_fixture(fixture)
    c = dict(meta["cfg"])
    if qwen:
        from crab_amd.unified_qwen import UnifiedConfig, UnifiedForCausalLM
        c.update(attention_bias=True)
    else:
        from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    c.update(num_hidden_layers=1, vocab_size=320, pad_token_id=2)
    model = get_peft_model(UnifiedForCausalLM(UnifiedConfig(**c), device="cuda"), LoraConfig())
    Wt = weights_from_table(meta)
    r = model.load_state_dict({"base_model.model." + k_: v for k_, v in Wt.items()}, strict=False)
    assert not r.unexpected_keys, r.unexpected_keys[:5]
    eng = model.base_model.model._engine
    x, xs = wide_layer_inputs(meta)
    B, S, D = x.shape
    M, steps = B * S, len(xs)
    Tmax = S + 8
    rows, step_seqs = A["rows"], A["step_seqs"]
    H, Hk = c["num_attention_heads"], c["num_key_value_heads"]
    # ---- the oracle on the same inputs: fp32 (must reproduce the fixture on EVERY sampled row), operand floor, storage emulation
    dev = _odev()
    ocfg = O.DecoderConfig(**{**meta["cfg"], "num_hidden_layers": 1, "vocab_size": 320})
    pos = torch.arange(S, device=dev)[None].expand(B, S)
    modes = {}
    for name, mode, W in (("fp32", None, Wt), ("floor", O.OPERANDS, Wt), ("emu", BF, stored_params(Wt))):
        Wd = {k_: v.to(dev) for k_, v in W.items()}
        cache = O.KVCache()
        y = O.decoder_layer(x.to(dev), Wd, 0, ocfg, cache, pos, emulate=mode).reshape(M, D)[rows.to(dev)].cpu()
        ys = [O.decoder_layer(x1.to(dev), Wd, 0, ocfg, cache, torch.full((B, 1), S + t, device=dev), emulate=mode)[step_seqs.to(dev), 0].cpu()
              for t, x1 in enumerate(xs)]
        modes[name] = (y, ys, cache.k[0][B - 1][meta["cache_heads"]].cpu(), cache.v[0][B - 1][meta["cache_heads"]].cpu())
        del Wd
    scale = A["y_rows"].abs().max().item()
    assert _err(modes["fp32"][0], A["y_rows"]) < 5e-4 and all(_err(a, b) < 5e-4 for a, b in zip(modes["fp32"][1], A["y_steps"])), "oracle (fp32, GPU) vs fixture"
    # ---- HIP: prefill of all B x S rows in one pass
    kc, vc = eng.alloc_cache(B, Tmax)
    with ops.launch_trace() as tr:
        eng.prefill(x.to(BF).cuda(), kc, vc, b0=0, all_logits=True)
    ws = eng._workspace(M)
    y_hip = ws.x[:M].float()[rows.cuda()].cpu()
    for k_, n in prefill_kernels.items():
        assert tr.launched(k_) == n, (k_, n, tr.counts)
    assert tr.launched("gemm_bt_glds_kernel") == 0 and tr.launched("qkv_rope_split_tile") == 0 and tr.launched("qkv_rope_split") == 0, tr.counts
    _bounded(f"{fixture}: layer output, prefill {B} x {S} rows vs the reference fixture", _err(y_hip, A["y_rows"]), _err(modes["floor"][0], A["y_rows"]),
             _err(modes["emu"][0], A["y_rows"]), scale, kernels=tr.counts)
    # ---- HIP: the cached one-token steps (position word on the device, RoPE + KV append in the q|k|v reduction)
    for t in range(steps):
        ops.cast_rows(xs[t][:, 0].to(BF).cuda().contiguous(), ws.x, B, D)
        posd = torch.full((1,), S + t, device="cuda", dtype=torch.int32)
        with ops.launch_trace() as tr:
            xo, _ = eng._layers(ws, B, 1, kc, vc, 0, Tmax, 0, posd, None)
        for k_ in decode_kernels:
            assert tr.launched(k_) >= 1, (k_, tr.counts)
        got = xo[:B].float()[step_seqs.cuda()].cpu()
        _bounded(f"{fixture}: layer output, cached decode step {t} ({B} rows) vs the reference fixture", _err(got, A["y_steps"][t]),
                 _err(modes["floor"][1][t], A["y_steps"][t]), _err(modes["emu"][1][t], A["y_steps"][t]), scale, kernels=tr.counts)
    # ---- cache rows of the last sequence (prefill rows + both appended rows): k after RoPE, v
    heads = meta["cache_heads"]
    for nm, cach, ref, fi in (("K", kc, A["cache_k"], 2), ("V", vc, A["cache_v"], 3)):
        got = cach[0, B - 1, heads, :S + steps].float().cpu()
        sc = ref.abs().max().item()
        # k / v are matrix OPERANDS: the floor rounds them once, so the HIP cache may not be further than 1.5 x that one rounding (+ the projection's own error)
        _bounded(f"{fixture}: {nm} cache rows (sequence {B - 1}) vs the reference fixture", _err(got, ref), _err(modes["floor"][fi], ref), _err(modes["emu"][fi], ref), sc)
    del model, eng, kc, vc
    torch.cuda.empty_cache()


def test_llama_layer_wide_vs_reference_fixture():
    """models/modeling_llama.py:352-465, 765-837 under peft_hyper/tuners/lora.py:260-369, 4096 / 11008 / 32 x 128."""
    _wide_layer("llama_layer_wide", False,
                prefill_kernels={"gemm_bt_ring_kernel+rope2": 1, "gemm_bt_ring_kernel": 3, "attn_fwd32_kernel<128,causal>": 1},
                decode_kernels=("attn_decode_kernel<128>", "splitk_epilogue_rope_kernel"))


def test_qwen_layer_wide_vs_reference_fixture():
    """models/qwen/modeling_qwen2.py:202-317, 712-809 with hyper-LoRA projections, 3584 / 18944 / GQA 28 : 4 x 128, q / k / v bias."""
    _wide_layer("qwen_layer_wide", True,
                prefill_kernels={"gemm_bt_ring_kernel+rope2": 1, "gemm_bt_ring_kernel": 3, "attn_fwd32_kernel<128,causal>": 1},
                decode_kernels=("attn_decode_gqa_kernel<128,7>", "splitk_epilogue_rope_kernel"))


@pytest.mark.parametrize("Bp,expect", [(1, ("gemm_skinny_dma_kernel", "rowfin_apply_kernel", "attn_decode_rope_kernel<128>")),
                                       (8, ("gemm_skinny_dma_kernel", "rowfin_apply_kernel", "attn_decode_rope_kernel<128,32>")),
                                       (48, ("gemm_bt_kernel(split-K)", "attn_decode_kernel<128>", "splitk_epilogue_rope_kernel")),
                                       (100, ("gemm_dec_ws_kernel", "attn_decode_kernel<128>", "splitk_epilogue_norm_kernel")),
                                       (256, ("gemm_dec_ws_kernel", "attn_decode_kernel<128>", "splitk_epilogue_norm_kernel")),
                                       (512, ("gemm_dec2_kernel", "attn_decode_kernel<128>", "splitk_epilogue_norm_kernel"))])
def test_llama_decode_regimes_vs_reference_fixture(Bp, expect):
    """EVERY decode regime of the projections against the SAME reference run (tests/golden/llama_decode_regimes_wide.npz: the Llama-2-7B-wide
    hyper-LoRA layer on 512 independent sequences of 8 prompt rows + one cached decode step): the first Bp sequences are prefilled and decoded
    as a Bp-row batch - 1 and 8 rows (the reference's own batch sizes: LDS-DMA skinny kernel, row-owning tails, fused small-batch attention), 48
    (64-row split-K kernels), 100 (the panel kernel above its 64-row floor, r06), 256 (panel kernel) and 512 rows (two-row-group panel kernel: the
    benchmark's regime) - and the stored sequences among them are compared with the reference's values; bound = 1.5 x max(operand floor,
    storage emulation) computed on those sequences; the launch trace proves the regime."""
    from crab_amd import ops
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    from oracle import crab_oracle as O
    from tests.util import stored_params, wide_layer_inputs
    meta, A = load_fixture("llama_decode_regimes_wide")
    c = dict(meta["cfg"])
    c.update(num_hidden_layers=1, vocab_size=320, pad_token_id=2)
    model = get_peft_model(UnifiedForCausalLM(UnifiedConfig(**c), device="cuda"), LoraConfig())
    Wt = weights_from_table(meta)
    r = model.load_state_dict({"base_model.model." + k_: v for k_, v in Wt.items()}, strict=False)
    assert not r.unexpected_keys
    eng = model.base_model.model._engine
    x, xs = wide_layer_inputs(meta)
    _, S, D = x.shape
    stored = A["seqs"].tolist()
    mine = [b for b in stored if b < Bp]
    at = [stored.index(b) for b in mine]
    assert len(mine) >= 1
    # ---- the oracle on the stored sequences (fp32 must reproduce the fixture; floor and emulation give the bound)
    dev = _odev()
    ocfg = O.DecoderConfig(**{**meta["cfg"], "num_hidden_layers": 1, "vocab_size": 320})
    runs = {}
    for name, mode, W in (("fp32", None, Wt), ("floor", O.OPERANDS, Wt), ("emu", BF, stored_params(Wt))):
        Wd = {k_: v.to(dev) for k_, v in W.items()}
        cache = O.KVCache()
        y = O.decoder_layer(x[mine].to(dev), Wd, 0, ocfg, cache, torch.arange(S, device=dev)[None].expand(len(mine), S), emulate=mode)[:, -1].cpu()
        y1 = O.decoder_layer(xs[0][mine].to(dev), Wd, 0, ocfg, cache, torch.full((len(mine), 1), S, device=dev), emulate=mode)[:, 0].cpu()
        runs[name] = (y, y1)
        del Wd
    assert _err(runs["fp32"][0], A["y_last"][at]) < 5e-4 and _err(runs["fp32"][1], A["y_step"][at]) < 5e-4
    # ---- HIP: prefill of the first Bp sequences, then ONE decode step of Bp rows on a decode workspace (what generate() decodes with)
    Tmax = 64
    kc, vc = eng.alloc_cache(Bp, Tmax)
    eng.prefill(x[:Bp].to(BF).cuda(), kc, vc, b0=0, all_logits=True)
    ws = eng._workspace(Bp * S)
    rows = torch.tensor([b * S + S - 1 for b in mine], device="cuda")
    y_hip = ws.x[:Bp * S].float()[rows].cpu()
    sc = A["y_last"].abs().max().item()
    _bounded(f"llama decode regimes: prefill of {Bp} x {S} rows, last rows vs the reference fixture", _err(y_hip, A["y_last"][at]), _err(runs["floor"][0], A["y_last"][at]),
             _err(runs["emu"][0], A["y_last"][at]), sc)
    wsd = eng._workspace(Bp, 0, decode=True)
    ops.cast_rows(xs[0][:Bp, 0].to(BF).cuda().contiguous(), wsd.x, Bp, D)
    posd = torch.full((1,), S, device="cuda", dtype=torch.int32)
    with ops.launch_trace() as tr:
        xo, _ = eng._layers(wsd, Bp, 1, kc, vc, 0, Tmax, 0, posd, None)
    for k_ in expect:
        assert tr.launched(k_) >= 1, (Bp, k_, tr.counts)
    got = xo[:Bp].float()[torch.tensor(mine, device="cuda")].cpu()
    _bounded(f"llama decode regimes: decode step of {Bp} rows vs the reference fixture", _err(got, A["y_step"][at]), _err(runs["floor"][1], A["y_step"][at]),
             _err(runs["emu"][1], A["y_step"][at]), A["y_step"].abs().max().item(), kernels=tr.counts)
    del model, eng, kc, vc
    torch.cuda.empty_cache()


def test_clip_wide_vs_reference_fixture():
    """CLIP ViT-L/14 widths, 36 frames x 257 tokens (M = 9252: every projection on the ring kernel, K = 1024 and 4096; head_dim 64 flash forward)."""
    from crab_amd import ops
    from crab_amd.multimodal_encoder import VisualEncoder
    from oracle import crab_oracle as O
    from tests.util import clip_wide_video, stored_params
    meta, A = load_fixture("clip_wide")
    W = weights_from_table(meta)
    ve = VisualEncoder(select_layer_list=meta["select"], config=meta["cfg"], device="cuda")
    r = ve.load_state_dict({k[len("model.visual_encoder."):]: v for k, v in W.items()}, strict=False)
    assert not r.unexpected_keys
    video = clip_wide_video(meta)
    with ops.launch_trace() as tr:
        feats = ve(ops.cast_bf16(video.cuda()))
    assert tr.launched("gemm_bt_ring_kernel") == 4 * max(meta["select"]) and tr.launched("attn_fwd32_kernel<64>") == max(meta["select"]), tr.counts
    cfg = O.ClipConfig(**meta["cfg"], select_layers=tuple(meta["select"]))
    dev = _odev()
    rows = A["rows"]
    outs = {}
    for name, mode, Wm in (("fp32", None, W), ("floor", O.OPERANDS, W), ("emu", BF, stored_params(W))):
        f = O.visual_encoder(video.to(dev), {k: v.to(dev) for k, v in Wm.items()}, cfg, emulate=mode)
        outs[name] = [t[0][rows.to(dev)].cpu() for t in f]
    for i in range(3):
        ref = A[f"f{i}"]
        assert _err(outs["fp32"][i], ref) < 5e-4
        _bounded(f"clip_wide level {i} (hidden state {meta['select'][i]}) vs the reference fixture", _err(feats[i][0][rows.cuda()], ref),
                 _err(outs["floor"][i], ref), _err(outs["emu"][i], ref), ref.abs().max().item(), kernels=tr.counts if i == 0 else None)


def test_beats_wide_vs_reference_fixture():
    """BEATs at 768 / 12 heads x 64 / 3072, two layers: 256 windows of L = 98 (n = 48; 12288 rows: the 768-wide GEMMs on the ring kernel) and 3 windows
    of L = 198 (n = 96), gated relative-position bias in both."""
    from crab_amd import ops
    from crab_amd.multimodal_encoder import AudioEncoder
    from oracle import crab_oracle as O
    from tests.util import beats_wide_audio, stored_params
    meta, A = load_fixture("beats_wide")
    W = weights_from_table(meta)
    ae = AudioEncoder(cfg=meta["cfg"], device="cuda")
    r = ae.load_state_dict({k[len("model.audio_encoder."):]: v for k, v in W.items()}, strict=False)
    assert not r.unexpected_keys and not r.missing_keys, r
    keys = O.BeatsConfig.__dataclass_fields__.keys()
    cfg = O.BeatsConfig(**{k: v for k, v in meta["cfg"].items() if k in keys})
    dev = _odev()
    for L in (98, 198):
        x = beats_wide_audio(meta, L)
        with ops.launch_trace() as tr:
            y = ae(ops.cast_bf16(x.cuda()))
        assert tr.launched("attn_fwd_kernel<64,bias>") == 2 and tr.launched("beats_gru_gate") == 2, tr.counts
        pick = (lambda t: t.reshape(-1, t.shape[-1])[A["rows98"].to(t.device)]) if L == 98 else (lambda t: t)
        if L == 98:
            assert tr.launched("gemm_bt_ring_kernel") >= 8, tr.counts          # q|k|v, out, fc1, fc2 of both layers at M = 12288
        ref = A[f"y{L}"]
        o = {}
        for name, mode, Wm in (("fp32", None, W), ("floor", O.OPERANDS, W), ("emu", BF, stored_params(W))):
            o[name] = pick(O.beats(x.to(dev), {k: v.to(dev) for k, v in Wm.items()}, cfg, emulate=mode)).cpu()
        assert _err(o["fp32"], ref) < 5e-4
        _bounded(f"beats_wide L={L} ({x.shape[0]} windows) vs the reference fixture", _err(pick(y), ref), _err(o["floor"], ref), _err(o["emu"], ref),
                 ref.abs().max().item(), kernels=tr.counts)


def test_projectors_wide_vs_reference_fixture():
    """Both Q-Former projectors at their real configuration: bert-base layers, 32 queries, cross-attention keys 1024 / 768 wide, MLP to 4096."""
    from crab_amd import ops
    from crab_amd.multimodal_encoder import ALProjector, VLProjector
    from oracle import crab_oracle as O
    from tests.util import bert_cfg, projectors_wide_inputs, stored_params
    meta, A = load_fixture("projectors_wide")
    W = weights_from_table(meta)
    bc = bert_cfg(meta["qf"])
    qf = O.QFormerConfig(hidden_size=meta["qf"]["hidden"], num_attention_heads=meta["qf"]["heads"], intermediate_size=meta["qf"]["inter"])
    vf, af = projectors_wide_inputs(meta)
    dev = _odev()
    vl = VLProjector(hidden_size=1024, image_token_nums=256, num_query_token=32, num_hidden_layers=2, d_model=meta["d_model"], depth=2, bert_config=bc,
                     device="cuda")
    r = vl.load_state_dict({k[len("model.vl_projector."):]: v for k, v in W.items() if k.startswith("model.vl_projector.")}, strict=False)
    assert not r.missing_keys, r.missing_keys
    al = ALProjector(hidden_size=768, num_query_token=32, num_hidden_layers=2, d_model=meta["d_model"], depth=2, bert_config=bc, device="cuda")
    r = al.load_state_dict({k[len("model.al_projector."):]: v for k, v in W.items() if k.startswith("model.al_projector.")}, strict=False)
    assert not r.missing_keys, r.missing_keys
    for name, mod, inp, fn, ref in (("VLProjector", vl, vf, O.vl_projector, A["vout"]), ("ALProjector", al, af, O.al_projector, A["aout"])):
        with ops.launch_trace() as tr:
            y = mod(inp.to(BF).cuda())
        o = {}
        for nm, mode, Wm in (("fp32", None, W), ("floor", O.OPERANDS, W), ("emu", BF, stored_params(W))):
            o[nm] = fn(inp.to(dev), {k: v.to(dev) for k, v in Wm.items()}, qf, emulate=mode).cpu()
        assert _err(o["fp32"], ref) < 5e-4
        _bounded(f"projectors_wide {name} vs the reference fixture", _err(y, ref), _err(o["floor"], ref), _err(o["emu"], ref), ref.abs().max().item(),
                 kernels=tr.counts)


def test_seg_module_wide_vs_reference_fixture():
    """SegModule under d_model 4096 / 1024-wide CLIP features / prompt dim 256 / 300 queries (models/multimodal_encoder.py:268-543, 891-1444)."""
    from crab_amd import ops
    from crab_amd.seg_module import SegModule
    from oracle import crab_oracle as O
    from tests.util import seg_wide_inputs, stored_params
    meta, A = load_fixture("seg_wide")
    W = weights_from_table(meta)
    seg = SegModule(d_model=meta["d_model"], vit_image_embedding_dim=meta["vit_dim"], device="cuda")
    seg.load_state_dict({k[len("model.seg_module."):]: v for k, v in W.items()}, strict=True)
    pred, feats = seg_wide_inputs(meta)
    with ops.launch_trace() as tr:
        out = seg(pred_embeddings=pred.to(BF).cuda(), multi_scale_image_feature_list=[f.to(BF).cuda() for f in feats], low_res_mask_size=112, gt_mask=None,
                  batch_task_names=meta["tasks"])['pred_masks']
    assert tuple(out[0].shape) == (71, 224, 224) and tuple(out[1].shape) == (1, 224, 224)
    emu = O.seg_module(pred, feats, meta["tasks"], stored_params(W))            # the oracle's SegModule has no emulation switch: bf16-rounded parameters, exact arithmetic
    fp = O.seg_module(pred, feats, meta["tasks"], W)
    subs = ((out[0][:, 3::8, 5::8], emu[0][:, 3::8, 5::8], fp[0][:, 3::8, 5::8], A["avss_sub"], "avss"),
            (out[1][:, 1::2, ::2], emu[1][:, 1::2, ::2], fp[1][:, 1::2, ::2], A["s4_sub"], "s4"))
    from tests.util import record_parity
    for got, e, f, ref, name in subs:
        assert _err(f, ref) < 5e-4
        hip, pe, sc = _err(got, ref), _err(e, ref), ref.abs().max().item()
        record_parity(f"seg_wide {name} masks vs the reference fixture", hip, sc, None, bf16_parameter_rounding_abs=pe, kernels=tr.counts if name == "avss" else None)
        # parameter rounding alone (weights in bf16, everything else exact) is a LOWER bound on what the bf16 path can reach here; the mask decoder's
        # activations add their own storage points: 3 x that bound (measured: see the parity report)
        assert hip <= 3.0 * pe, (name, hip, pe, sc)
