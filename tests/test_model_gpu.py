"""End-to-end GPU parity: the product modules (crab_amd.*, HIP kernels through the C-ABI) against
  (1) the reference-generated golden fixtures (fp32 reference outputs), and
  (2) the oracle on the same seeded weights, in fp32 and in bf16-storage emulation.

Tolerances (r06): north_star asks for "logits within 1e-3 bf16"; what bf16 MATRIX OPERANDS allow is computed, not asserted from memory: for
every component the oracle is executed as the bf16-operand floor and as the bf16-storage emulation on the same fixture (tests/bounds.py over
scripts/parity_floor.py, ~10 s of CPU per session), and the HIP path must stay within 1.5 x the larger of the two - measured against the SAME
fp32 reference values.  The floors are 2.9e-3 ... 6.9e-3 on these stacks (tests/test_parity_floor.py pins "> 1e-3" on the CPU).
  * greedy token ids: exact wherever the fp32 reference's top-2 logit margin exceeds twice the measured logit error; on the sharpened-logit
    fixture (tests/golden/sharp_tiny_llama.npz, margins >= 10 x the bf16 error) exact on every step, no escape.
"""
import pytest
import torch

from tests.util import build_tiny_crab, load_fixture, weights_from_table, bert_cfg

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
# r06: NO tolerance constants.  Every bound below is COMPUTED in the test session from the oracle (tests/bounds.py over scripts/parity_floor.py):
#     hip  <=  1.5 x max(bf16-operand floor, bf16-storage emulation)  of the component under test,
# both measured against the same fp32 reference the HIP path is compared with (PB.bound(<component>)); a comparison with the storage emulation
# itself gets 2.5 x (two executions within that distance of fp32), HIP-vs-HIP comparisons of two kernel regimes 2 x the component's bound.
from tests import bounds as PB


def _rel(got, ref, what=""):
    from tests.util import rel_err
    return rel_err(got, ref, what)


def _bf(W):
    from tests.util import stored_params
    return stored_params(W)             # bf16 everywhere except the encoders' LayerNorm parameters (fp32 in the HIP modules since r05)


def test_clip_tower_vs_reference_fixture_and_oracle():
    from crab_amd import synth
    from crab_amd.multimodal_encoder import VisualEncoder
    from oracle import crab_oracle as O
    meta, A = load_fixture("clip_tiny")
    W = weights_from_table(meta)
    ve = VisualEncoder(select_layer_list=meta["select"], config=meta["cfg"], device="cuda")
    missing = ve.load_state_dict({k[len("model.visual_encoder."):]: v for k, v in W.items()}, strict=False)
    assert not missing.unexpected_keys
    video = synth.synth_video(meta["t_v"], seed=meta["seed"], clip=meta["clip"])[None]
    from crab_amd import ops
    feats = ve(ops.cast_bf16(video.cuda()))
    cfg = O.ClipConfig(**meta["cfg"], select_layers=tuple(meta["select"]))
    emu = O.visual_encoder(video.to(BF).float(), _bf(W), cfg, emulate=BF)
    for i in range(3):
        assert _rel(feats[i], A[f"f{i}"], f"clip_tiny level {i} vs fp32 reference") < PB.bound(f"clip_tiny feature levels [{i}]"), f"level {i} vs reference"
        assert _rel(feats[i], emu[i], f"clip_tiny level {i} vs bf16-emulating oracle") < PB.bound(f"clip_tiny feature levels [{i}]", PB.FACTOR_VS_EMULATION), f"level {i} vs bf16-emulating oracle"


def test_beats_vs_reference_fixture_and_oracle():
    from crab_amd import ops
    from crab_amd.multimodal_encoder import AudioEncoder
    from oracle import crab_oracle as O
    meta, A = load_fixture("beats_tiny")
    W = weights_from_table(meta)
    ae = AudioEncoder(cfg=meta["cfg"], device="cuda")
    r = ae.load_state_dict({k[len("model.audio_encoder."):]: v for k, v in W.items()}, strict=False)
    assert not r.unexpected_keys and not r.missing_keys, r
    keys = O.BeatsConfig.__dataclass_fields__.keys()
    cfg = O.BeatsConfig(**{k: v for k, v in meta["cfg"].items() if k in keys})
    for L in (98, 198):
        x = A[f"x{L}"]
        y = ae(ops.cast_bf16(x.cuda()))
        assert _rel(y, A[f"y{L}"], f"beats_tiny L={L} vs fp32 reference") < PB.bound(f"beats_tiny L={L}"), f"L={L} vs reference"
        assert _rel(y, O.beats(x.to(BF).float(), _bf(W), cfg, emulate=BF), f"beats_tiny L={L} vs bf16-emulating oracle") < PB.bound(f"beats_tiny L={L}", PB.FACTOR_VS_EMULATION), f"L={L} vs emulating oracle"


def test_projectors_vs_reference_fixture():
    from crab_amd.multimodal_encoder import ALProjector, VLProjector
    meta, A = load_fixture("projectors_tiny")
    W = weights_from_table(meta)
    bc = bert_cfg(meta["qf"])
    vl = VLProjector(hidden_size=128, image_token_nums=256, num_query_token=32, num_hidden_layers=2, d_model=meta["d_model"],
                     depth=2, bert_config=bc, device="cuda")
    r = vl.load_state_dict({k[len("model.vl_projector."):]: v for k, v in W.items() if k.startswith("model.vl_projector.")}, strict=False)
    assert not r.missing_keys, r.missing_keys
    assert _rel(vl(A["vfeat"].to(BF).cuda()), A["vout"], "VLProjector vs fp32 reference") < PB.bound("VLProjector")
    al = ALProjector(hidden_size=128, num_query_token=32, num_hidden_layers=2, d_model=meta["d_model"], depth=2, bert_config=bc,
                     device="cuda")
    r = al.load_state_dict({k[len("model.al_projector."):]: v for k, v in W.items() if k.startswith("model.al_projector.")}, strict=False)
    assert not r.missing_keys, r.missing_keys
    assert _rel(al(A["afeat"].to(BF).cuda()), A["aout"], "ALProjector vs fp32 reference") < PB.bound("ALProjector")


def test_native_encoder_layer_sequencers_equal_python_sequences():
    """crab_clip_layer / crab_beats_layer / crab_qformer_layer (csrc/encoder_layers.hip) against the per-launch Python sequences of
    crab_amd/multimodal_encoder.py on the tiny fixtures: CLIP hidden states at the three selected levels, BEATs features (L = 98 and
    198), both projectors - bit-identical (same launches in the same order)."""
    from crab_amd import multimodal_encoder as ME, ops, synth
    meta, A = load_fixture("clip_tiny")
    W = weights_from_table(meta)
    ve = ME.VisualEncoder(select_layer_list=meta["select"], config=meta["cfg"], device="cuda")
    ve.load_state_dict({k[len("model.visual_encoder."):]: v for k, v in W.items()}, strict=False)
    video = ops.cast_bf16(synth.synth_video(meta["t_v"], seed=meta["seed"], clip=meta["clip"])[None].cuda())
    meta_b, Ab = load_fixture("beats_tiny")
    ae = ME.AudioEncoder(cfg=meta_b["cfg"], device="cuda")
    ae.load_state_dict({k[len("model.audio_encoder."):]: v for k, v in weights_from_table(meta_b).items()}, strict=False)
    meta_p, Ap = load_fixture("projectors_tiny")
    Wp = weights_from_table(meta_p)
    bc = bert_cfg(meta_p["qf"])
    vl = ME.VLProjector(hidden_size=128, image_token_nums=256, num_query_token=32, num_hidden_layers=2, d_model=meta_p["d_model"], depth=2,
                        bert_config=bc, device="cuda")
    vl.load_state_dict({k[len("model.vl_projector."):]: v for k, v in Wp.items() if k.startswith("model.vl_projector.")}, strict=False)
    al = ME.ALProjector(hidden_size=128, num_query_token=32, num_hidden_layers=2, d_model=meta_p["d_model"], depth=2, bert_config=bc, device="cuda")
    al.load_state_dict({k[len("model.al_projector."):]: v for k, v in Wp.items() if k.startswith("model.al_projector.")}, strict=False)
    outs = []
    for native in (True, False):
        ME.NATIVE_ENC_LAYERS = native
        try:
            r = [f.clone() for f in ve(video)]
            r += [ae(ops.cast_bf16(Ab[f"x{L}"].cuda())).clone() for L in (98, 198)]
            r += [vl(Ap["vfeat"].to(BF).cuda()).clone(), al(Ap["afeat"].to(BF).cuda()).clone()]
            outs.append(r)
        finally:
            ME.NATIVE_ENC_LAYERS = True
    assert len(outs[0]) == len(outs[1]) == 7
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)


def _inputs(meta):
    from crab_amd import synth
    p = meta["prompts"]
    return [{'<video>': synth.synth_video(p["t_v"], seed=meta["seed"], clip=c),
             '<audio>': synth.synth_audio(p["t_a"], p["l_a"], seed=meta["seed"], clip=c)} for c in (p["clip0"], p["clip1"])]


def _check_ids(ids, ref_ids, ref_logits, got_logits, min_frac=1.0):
    """ids must agree up to (and including) every step whose reference top-2 margin exceeds 2x the measured
    max logit error; after a sub-margin step the sequences may legitimately diverge.  `min_frac` = the fraction of steps that
    must be covered before such a divergence: 1.0 (every step of every row, what MI355X measured on every fixture -
    profiles/r02_parity_report.json: steps_checked == steps_total in all 20 rows) unless a caller states a measured lower value."""
    ids = ids.cpu()
    got_logits = got_logits.float().cpu()
    top2 = ref_logits.topk(2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    worst, checked, total = 0.0, 0, 0
    for b in range(ref_ids.shape[0]):
        for s in range(ref_ids.shape[1]):
            total += 1
            # contexts are identical up to here, so the logits are comparable
            err = (got_logits[b, s] - ref_logits[b, s]).abs().max().item()
            worst = max(worst, err)
            if ids[b, s] != ref_ids[b, s]:
                assert margin[b, s] <= 2 * err, (f"row {b} step {s}: id {ids[b, s]} vs {ref_ids[b, s]} although the reference margin "
                                                 f"{margin[b, s]:.4f} exceeds twice the logit error {err:.4f}")
                break          # legitimately diverged at a sub-noise margin: later steps see different contexts
            checked += 1
    assert checked >= min_frac * total, f"greedy-id parity covered only {checked}/{total} steps (required fraction {min_frac})"
    from tests.util import record_parity
    record_parity("greedy per-step last-row logits, worst step", worst, ref_logits.abs().max().item(), None, steps_checked=checked, steps_total=total,
                  min_ref_margin=float(margin.min()))
    return worst


def test_forward_honours_left_pad_mask_and_position_ids_like_the_reference():
    """forward() (models/unified_llama.py:129-160) hands prepare_multimodal_inputs' attention_mask and cumsum-1 position_ids to the
    decoder - generate() drops them.  Reference-recorded logits of the left-padded bs-2 batch (valid rows; a pad row sees no key and
    is undefined in the reference), the kept cache, then the 1-token decode shortcut (:125-127) with the extended mask and per-row
    positions; finally the multimodal branch itself (batch_input_ids=...), which must route mask and positions the same way."""
    meta, A = load_fixture("forward_masked_tiny_llama")
    W = weights_from_table(meta)
    model = build_tiny_crab(meta)
    r = model.load_state_dict(W, strict=False)
    assert not r.missing_keys, r.missing_keys[:5]
    um = model.base_model.model
    mask, pos = A["mask_bs2"], A["pos_bs2"]
    valid = mask.bool()
    out = um(inputs_embeds=A["embeds_bs2"].cuda(), attention_mask=mask.cuda(), position_ids=pos.cuda(), use_cache=True, output_hidden_states=True)
    assert torch.isfinite(out.logits).all()
    assert _rel(out.logits.cpu()[valid], A["logits_bs2"][valid], "forward() with left-pad mask + position_ids: logits of valid rows vs fp32 reference") < PB.bound("forward_masked_tiny_llama: left-pad")
    assert _rel(out.hidden_states[-1].float().cpu()[valid], A["hidden_bs2"][valid], "forward() with left-pad mask: post-norm hidden of valid rows") < PB.bound("forward_masked_tiny_llama: left-pad")
    # without the mask the padded row is far off (the reference: 3.3 on a logit scale of 3.8) - the mask path is really exercised
    plain = um(inputs_embeds=A["embeds_bs2"].cuda())
    assert float((plain.logits.cpu()[1] - A["logits_bs2"][1])[valid[1]].abs().max()) > 0.5
    step = um(input_ids=A["step_tok"][:, None].cuda(), attention_mask=A["step_mask"].cuda(), position_ids=A["step_pos"].cuda(),
              past_key_values=out.past_key_values)
    assert _rel(step.logits.cpu(), A["step_logits"], "forward() decode shortcut with extended mask + per-row positions vs fp32 reference") < PB.bound("forward_masked_tiny_llama: left-pad")
    # the multimodal branch: encoders -> splice -> left pad -> decoder with mask / positions
    mods = _inputs(meta)
    lab = [torch.full_like(A["ids0"], -100), torch.full_like(A["ids1"], -100)]
    mm = um(batch_input_ids=[A["ids0"], A["ids1"]], batch_labels=lab, batch_X_modals=mods, batch_task_names=['avqa', 'avqa'])
    assert _rel(mm.logits.cpu()[valid], A["logits_bs2"][valid], "forward(batch_input_ids=...) left-padded bs 2 vs fp32 reference") < PB.bound("multimodal forward_masked_tiny_llama")


def test_forward_accepts_any_2d_attention_mask_like_the_reference():
    """HF's mask utilities accept any 2-D attention_mask (padding mask AND causal mask), not only left padding: reference-recorded logits of
    the hyper-LoRA tiny Llama under a mask with interior holes, a masked last key and left pads - default (arange) positions, then explicit
    cumsum-1 position_ids - the kept cache, and the 1-token decode shortcut (models/unified_llama.py:125-127) with the extended mask.  The
    mask travels as one visibility bit per key (crab_attn_desc.key_mask, crab_attn_decode_keymask).  Query rows that see no key at all
    (the left pads) are undefined in the reference and excluded."""
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    meta, A = load_fixture("forward_holes_tiny_llama")
    W = weights_from_table(meta)
    cfg = UnifiedConfig(**meta["dec"], pad_token_id=2)
    model = get_peft_model(UnifiedForCausalLM(cfg, device="cuda"), LoraConfig())
    r = model.load_state_dict(W, strict=False)
    assert not r.missing_keys and not r.unexpected_keys, r
    um = model.base_model.model
    mask = A["mask"]
    seen = mask.cumsum(-1) > 0
    emb = A["embeds"].cuda()
    out = um(inputs_embeds=emb, attention_mask=mask.cuda(), use_cache=True, output_hidden_states=True)
    assert torch.isfinite(out.logits).all()
    assert _rel(out.logits.cpu()[seen], A["logits"][seen], "forward() under a mask with interior holes: logits of defined rows vs fp32 reference") < PB.bound("forward_holes_tiny_llama")
    assert _rel(out.hidden_states[-1].float().cpu()[seen], A["hidden"][seen], "forward() under a mask with holes: post-norm hidden") < PB.bound("forward_holes_tiny_llama")
    plain = um(inputs_embeds=emb)
    assert float((plain.logits.cpu() - A["logits"])[seen].abs().max()) > 0.5          # the mask path is really exercised
    outp = um(inputs_embeds=emb, attention_mask=mask.cuda(), position_ids=A["pos"].cuda())
    assert _rel(outp.logits.cpu()[seen], A["logits_pos"][seen], "forward() under a mask with holes + cumsum-1 position_ids") < PB.bound("forward_holes_tiny_llama")
    step = um(input_ids=A["step_tok"][:, None].cuda(), attention_mask=A["step_mask"].cuda(), position_ids=A["step_pos"].cuda(),
              past_key_values=out.past_key_values)
    assert _rel(step.logits.cpu(), A["step_logits"], "forward() decode shortcut over a cache with masked rows vs fp32 reference") < PB.bound("forward_holes_tiny_llama")
    # the bit mask and the first-visible-key form are two encodings of the same thing for a left-padded batch: identical logits
    pad = torch.ones_like(mask); pad[1, :5] = 0
    a = um(inputs_embeds=emb, attention_mask=pad.cuda()).logits
    ks, km = um._key_visibility(pad, 2, mask.shape[1], mask.shape[1])
    assert km is None and ks.tolist() == [0, 5]
    from crab_amd import ops
    eng = um._engine
    kc, vc = eng.alloc_cache(2, 64)
    b_, _ = eng.prefill(emb.to(BF), kc, vc, all_logits=True, key_mask=ops.pack_key_mask(pad).cuda())
    assert torch.equal(a[pad.bool()], b_[pad.bool()])


@pytest.mark.parametrize("fixture", ["full_tiny_llama", "full_tiny_qwen"])
def test_full_tiny_generate_matches_reference(fixture):
    """encoders -> prepare_multimodal_inputs -> hyper-LoRA decoder -> greedy ids against the reference-recorded fixture:
    full_tiny_llama = models/unified_llama.py (BASELINE configs[1]); full_tiny_qwen = the class the reference's eval script selects,
    models/unified_qwen.py over Qwen2 (GQA 4/2, q/k/v bias, projector width 256 != encoder width; BASELINE configs[2])."""
    meta, A = load_fixture(fixture)
    W = weights_from_table(meta)
    model = build_tiny_crab(meta)
    r = model.load_state_dict(W, strict=False)
    assert not r.missing_keys, r.missing_keys[:5]
    assert model.SPECIAL_TOKEN_2_IDS == meta["special"]
    mods = _inputs(meta)
    lab = [torch.full_like(A["ids0"], -100), torch.full_like(A["ids1"], -100)]
    inp1 = model.prepare_multimodal_inputs([A["ids0"]], [lab[0]], [mods[0]], ['avqa'])
    assert _rel(inp1["inputs_embeds"], A["embeds_bs1"], f"{fixture}: inputs_embeds bs1 (encoders + projectors + splice)") < PB.bound(f"{fixture}: inputs_embeds")
    inp2 = model.prepare_multimodal_inputs([A["ids0"], A["ids1"]], lab, mods, ['avqa', 'avqa'])
    assert _rel(inp2["inputs_embeds"], A["embeds_bs2"], f"{fixture}: inputs_embeds left-padded bs2") < PB.bound(f"{fixture}: inputs_embeds")
    assert torch.equal(inp2["position_ids"].cpu().long(), A["pos_bs2"].long())
    assert torch.equal(inp2["attention_mask"].cpu().long(), A["mask_bs2"].long())
    # prefill, all rows (LlamaForCausalLM.forward)
    out = model.base_model.model(inputs_embeds=A["embeds_bs1"].to(BF).cuda(), output_hidden_states=True)
    assert _rel(out.logits, A["prefill_logits_bs1"], f"{fixture}: prefill logits, all rows") < PB.bound(f"{fixture}: decoder prefill logits")
    assert _rel(out.hidden_states[-1], A["prefill_hidden_bs1"], f"{fixture}: post-norm hidden, all rows") < PB.bound(f"{fixture}: decoder prefill logits")
    # generate: public API, bs=1 and left-padded bs=2, graph replay and plain launches must agree bit for bit
    n = meta["new_tokens"]
    kw = dict(use_cache=True, max_new_tokens=n, do_sample=False, pad_token_id=2, eos_token_id=None,
              output_logits=True, return_dict_in_generate=True)
    for bs, key in ((1, "bs1"), (2, "bs2")):
        bi = [A["ids0"], A["ids1"]][:bs]
        r1 = model.generate(batch_input_ids=bi, batch_labels=lab[:bs], batch_X_modals=mods[:bs], batch_task_names=['avqa'] * bs, **kw)
        r2 = model.generate(batch_input_ids=bi, batch_labels=lab[:bs], batch_X_modals=mods[:bs], batch_task_names=['avqa'] * bs,
                            use_graph=False, **kw)
        assert torch.equal(r1.sequences, r2.sequences), "HIP-graph replay differs from plain launches"
        got = torch.stack(r1.logits, 1)
        assert torch.equal(got, torch.stack(r2.logits, 1))
        err = _check_ids(r1.sequences, A[f"ids_{key}"], A[f"logits_{key}"], got)
        assert err < PB.bound(f"{fixture}: end to end") * A[f"logits_{key}"].abs().max().item(), err
        plain = model.generate(batch_input_ids=bi, batch_labels=lab[:bs], batch_X_modals=mods[:bs], batch_task_names=['avqa'] * bs,
                               use_cache=True, max_new_tokens=n, pad_token_id=2, eos_token_id=None)
        assert torch.equal(plain, r1.sequences) and plain.shape == (bs, n)


def test_tiny_qwen2_gqa_bias_generate_matches_reference():
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_qwen import UnifiedConfig, UnifiedForCausalLM
    meta, A = load_fixture("decoder_tiny_qwen2")
    W = weights_from_table(meta)
    cfg = UnifiedConfig(**meta["dec"], attention_bias=True, pad_token_id=2)
    model = get_peft_model(UnifiedForCausalLM(cfg, device="cuda"), LoraConfig())
    r = model.load_state_dict(W, strict=False)
    assert not r.missing_keys and not r.unexpected_keys, r
    res = model.generate(inputs_embeds=A["embeds"].to(BF).cuda(), max_new_tokens=meta["new_tokens"], pad_token_id=2,
                         eos_token_id=None, output_logits=True, return_dict_in_generate=True)
    got = torch.stack(res.logits, 1)
    err = _check_ids(res.sequences, A["ids"], A["logits"], got)
    from oracle import crab_oracle as O
    assert err < PB.decoder_bound(A["embeds"].to(BF), O.strip_peft_prefix(W), O.DecoderConfig(**meta["dec"]), A["ids"]) * A["logits"].abs().max().item(), err


def test_eos_and_min_new_tokens_semantics():
    """Finished rows emit pad, generation stops when every row hit EOS, min_new_tokens suppresses EOS."""
    meta, A = load_fixture("full_tiny_llama")
    W = weights_from_table(meta)
    model = build_tiny_crab(meta)
    model.load_state_dict(W, strict=False)
    emb = A["embeds_bs2"].to(BF).cuda()
    ref_ids = A["ids_bs2"]
    eos = int(ref_ids[0, 3])                      # make the 4th token of row 0 the EOS
    ids = model.generate(inputs_embeds=emb, max_new_tokens=12, eos_token_id=eos, pad_token_id=2).cpu()
    # expected from the reference's greedy ids: rows are independent, a finished row emits pad, stop when all finished
    exp = ref_ids.clone()
    done_at = []
    for b in range(exp.shape[0]):
        hit = (exp[b] == eos).nonzero()
        k = int(hit[0]) if hit.numel() else exp.shape[1] - 1
        exp[b, k + 1:] = 2
        done_at.append(k if hit.numel() else exp.shape[1] - 1)
    exp = exp[:, : max(done_at) + 1]
    assert ids.shape == exp.shape and torch.equal(ids, exp), (ids.tolist(), exp.tolist())
    ids2 = model.generate(inputs_embeds=emb, max_new_tokens=12, eos_token_id=eos, pad_token_id=2, min_new_tokens=12)
    assert ids2.shape[1] == 12 and not (ids2 == eos).any()
    # decode groups on separate HIP streams (one graph each): rows never interact, same ids and same EOS trimming
    ids3 = model.generate(inputs_embeds=emb, max_new_tokens=12, eos_token_id=eos, pad_token_id=2, decode_streams=2).cpu()
    assert ids3.shape == exp.shape and torch.equal(ids3, exp), (ids3.tolist(), exp.tolist())
    ids4 = model.generate(inputs_embeds=emb, max_new_tokens=12, eos_token_id=eos, pad_token_id=2, min_new_tokens=12, decode_streams=2)
    assert torch.equal(ids4, ids2)


def test_seg_module_vs_reference_fixture_and_oracle():
    """SegModule (generate_avs pixel path, BASELINE config 5) at the fixture configuration: avss (71 classes) and s4."""
    from crab_amd import synth
    from crab_amd.seg_module import SegModule
    from oracle import crab_oracle as O
    from tests.util import seg_inputs
    meta, A = load_fixture("seg_tiny")
    W = weights_from_table(meta)
    seg = SegModule(d_model=meta["d_model"], vit_image_embedding_dim=128, device="cuda")
    r = seg.load_state_dict({k[len("model.seg_module."):]: v for k, v in W.items()}, strict=True)
    pred, feats = seg_inputs(meta)
    out = seg(pred_embeddings=pred.to(BF).cuda(), multi_scale_image_feature_list=[f.to(BF).cuda() for f in feats],
              low_res_mask_size=112, gt_mask=None, batch_task_names=meta["tasks"])['pred_masks']
    assert tuple(out[0].shape) == (71, 224, 224) and tuple(out[1].shape) == (1, 224, 224)
    ref = O.seg_module(pred.to(BF).float(), [f.to(BF).float() for f in feats], meta["tasks"], _bf(W))
    for i in range(2):
        assert _rel(out[i], ref[i], f"SegModule sample {i} vs oracle on bf16-rounded weights") < 1.3e-2, f"sample {i} vs oracle on bf16-rounded weights"
    assert _rel(out[0][:, 3::8, 5::8], A["avss_sub"], "SegModule avss vs fp32 reference") < 2.1e-2          # r05 (fp32 LayerNorm parameters): 1.15e-2 / 1.39e-2 measured (r04: 1.36e-2 / 1.77e-2 under 3.2e-2)
    assert _rel(out[1][:, 1::2, ::2], A["s4_sub"], "SegModule s4 vs fp32 reference") < 2.1e-2


def _avs_setup():
    """Tiny Crab with the SegModule; the six <mask_i> ids re-pointed at the tokens the random decoder emits at steps 1..6."""
    from crab_amd import synth
    from oracle import crab_oracle as O
    from tests.test_oracle_golden import _full_cfg
    from tests.util import DuckTokenizer, bert_cfg
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    meta, A = load_fixture("full_tiny_llama")
    smeta, _ = load_fixture("seg_tiny")
    W = weights_from_table(meta)
    Wseg = weights_from_table(smeta)
    W.update({"base_model.model." + k: v for k, v in Wseg.items()})
    cfg = UnifiedConfig(**meta["dec"], pad_token_id=meta["pad_token_id"])
    cfg.vocab_size = meta["base_vocab"]
    model = get_peft_model(UnifiedForCausalLM(cfg, device="cuda"), LoraConfig())
    model.get_model().pad_token_id = meta["pad_token_id"]
    model.get_model().init_multimodal_modules(d_model=meta["d_model"], visual_branch=True, audio_branch=True, segment_branch=True,
                                              select_layer_list=meta["select"], clip_config=meta["clip"], beats_config=meta["beats"],
                                              bert_config=bert_cfg(meta["qf"]), vit_image_embedding_dim=meta["clip"]["hidden_size"])
    model.initialize_MM_tokenizer(DuckTokenizer(meta["base_vocab"]), mask_token_nums=6)
    r = model.load_state_dict(W, strict=False)
    assert not r.missing_keys, r.missing_keys[:5]
    sp = model.SPECIAL_TOKEN_2_IDS
    ids = A["ids0"].clone()
    for a_, b_ in (("<video_start>", "<image_start>"), ("<video>", "<image>"), ("<video_end>", "<image_end>")):
        ids[ids == sp[a_]] = sp[b_]
    p = meta["prompts"]
    image = synth.synth_video(1, seed=meta["seed"], clip=p["clip0"])
    mods = [{'<image>': image, '<audio>': synth.synth_audio(p["t_a"], p["l_a"], seed=meta["seed"], clip=p["clip0"])}]
    lab = [torch.full_like(ids, -100)]
    n = 8
    plain = model.generate(batch_input_ids=[ids], batch_labels=lab, batch_X_modals=mods, batch_task_names=['s4'], max_new_tokens=n,
                           pad_token_id=2, eos_token_id=None).cpu()
    for i in range(6):
        sp[f'<mask_{i}>'] = int(plain[0, 1 + i])
    return model, meta, A, W, sp, ids, image, mods, lab, n, plain


def test_generate_avs_pipeline_vs_oracle():
    """generate_avs (unified_llama.py:270-361): <image> prompt -> greedy ids + per-step hidden states -> the states of the
    steps followed by a <mask_i> token -> SegModule masks.  The tiny random decoder never emits real mask tokens, so the six
    <mask_i> ids are re-pointed at the tokens it does emit at steps 1..6 (the selection logic is what is under test)."""
    from oracle import crab_oracle as O
    from tests.test_oracle_golden import _full_cfg
    model, meta, A, W, sp, ids, image, mods, lab, n, plain = _avs_setup()
    res = model.generate_avs(batch_input_ids=[ids], batch_labels=lab, batch_X_modals=mods, batch_task_names=['s4'], max_new_tokens=n,
                             pad_token_id=2, eos_token_id=None)
    assert torch.equal(res['output_ids'].cpu(), plain)
    assert len(res['pred_masks']) == 1 and tuple(res['pred_masks'][0].shape) == (1, 224, 224)
    # oracle pipeline on bf16-rounded weights
    Wo = _bf(O.strip_peft_prefix(W))
    ocfg = _full_cfg(meta)
    inp = O.prepare_multimodal_inputs([ids], mods, Wo, ocfg)
    oids, _, ohid = O.greedy_generate(inp["inputs_embeds"], Wo, ocfg.decoder, n, pad_token_id=2, return_hidden=True)
    row = plain[0].tolist()
    seg_ids = {sp[f'<mask_{i}>'] for i in range(6)}
    picks = [j for j in range(n - 1) if row[j + 1] in seg_ids][-6:]
    assert len(picks) == 6
    feats = O.visual_encoder(image[None].to(BF).float(), Wo, ocfg.clip)
    ref = O.seg_module(torch.stack([ohid[:, j] for j in picks], 1), feats[:2], ['s4'], Wo)
    if not torch.equal(oids, plain):
        # the ids may only diverge at a step whose oracle top-2 margin is below the bf16 logit noise; a super-margin divergence is a
        # failure, not a reason to skip the mask comparison
        _, olog, _ = O.greedy_generate(inp["inputs_embeds"], Wo, ocfg.decoder, n, pad_token_id=2, return_hidden=True)
        j = int((oids[0] != plain[0]).nonzero()[0])
        top2 = olog[0, j].topk(2).values
        assert float(top2[0] - top2[1]) < 0.1 * float(olog.abs().max()), f"generate_avs ids diverge from the oracle at step {j} at a super-margin step"
        pytest.skip(f"ids diverge at sub-margin step {j}: masks are not comparable")
    else:
        assert _rel(res['pred_masks'][0], ref[0], "generate_avs masks vs oracle pipeline") < 1.3e-2      # measured 8.5e-3


def _avs_samples(model, meta, sp, ids, image, mods, n):
    """Seven one-sample calls of the pixel loops: the margin-checked sample under three tasks (two class counts), the same clip behind a longer
    prompt, and other clips (whose greedy ids do not hit the six re-pointed <mask_i> ids: the ids-only outcome)."""
    from crab_amd import synth
    p = meta["prompts"]
    mk = lambda ids_, mods_, task: {"batch_input_ids": [ids_], "batch_labels": [torch.full_like(ids_, -100)], "batch_X_modals": [mods_], "batch_task_names": [task]}
    longer = torch.cat([ids[:2], torch.tensor([7, 9, 11, 13, 15]), ids[2:]])
    samples = [mk(ids, mods[0], 's4'), mk(ids, mods[0], 'avss'), mk(longer, mods[0], 'ms3'), mk(ids, mods[0], 'ms3')]
    for c in (31, 32, 33):
        m2 = {'<image>': synth.synth_video(1, seed=meta["seed"], clip=c), '<audio>': synth.synth_audio(p["t_a"], p["l_a"], seed=meta["seed"], clip=c)}
        samples.append(mk(ids, m2, 'avss' if c == 32 else 's4'))
    return samples


def test_generate_avs_many_equals_one_sample_calls():
    """generate_avs_many / generate_avs(bs > 1) (r06): several calls of the pixel loops executed together - one ragged decode batch with per-step
    hidden states, per-row <mask_i> picks, the SegModule batched per class count - against the SAME calls made one by one (the reference's loops,
    scripts/quick_start.py:270-450): ids equal (up to a sub-margin step of the coalesced decode kernels), masks within the mask decoder's tolerance,
    the ids-only outcome for rows without six mask tokens, mixed class counts in one batch."""
    model, meta, A, W, sp, ids, image, mods, lab, n, plain = _avs_setup()
    samples = _avs_samples(model, meta, sp, ids, image, mods, n)
    kw = dict(max_new_tokens=n, pad_token_id=2, eos_token_id=None)
    one = [model.generate_avs(**s_, **kw) for s_ in samples]
    many = model.generate_avs_many(samples, **kw)
    assert len(many) == len(samples)
    n_masks = 0
    for i, (a, b) in enumerate(zip(one, many)):
        if not torch.equal(a['output_ids'], b['output_ids']):
            # the coalesced wave runs the decode kernels of ITS row count: a step may flip only where the one-sample call's own top-2 margin is tiny
            lg = model.generate(**samples[i], **kw, output_logits=True, return_dict_in_generate=True)
            j = int((a['output_ids'][0] != b['output_ids'][0]).nonzero()[0])
            top2 = lg.logits[j][0].float().topk(2).values
            assert float(top2[0] - top2[1]) < 0.05 * float(lg.logits[j].abs().max()), (i, j)
            continue
        assert ('pred_masks' in a) == ('pred_masks' in b), i
        if 'pred_masks' in a:
            n_masks += 1
            assert a['pred_masks'][0].shape == b['pred_masks'][0].shape == ((71, 224, 224) if samples[i]["batch_task_names"][0] == 'avss' else (1, 224, 224))
            assert _rel(b['pred_masks'][0], a['pred_masks'][0], f"generate_avs_many sample {i} ({samples[i]['batch_task_names'][0]}) masks vs its own generate_avs call (HIP vs HIP)") < 1.3e-2
    assert n_masks >= 3 and any('pred_masks' not in a for a in one), "the sample set must cover both outcomes"
    # the public bs > 1 form: ids padded to one tensor, masks as a list with None for the ids-only rows
    merged = model.generate_avs(batch_input_ids=[s_["batch_input_ids"][0] for s_ in samples], batch_labels=[s_["batch_labels"][0] for s_ in samples],
                                batch_X_modals=[s_["batch_X_modals"][0] for s_ in samples], batch_task_names=[s_["batch_task_names"][0] for s_ in samples], **kw)
    assert merged['output_ids'].shape == (len(samples), n) and len(merged['pred_masks']) == len(samples)
    for i, b in enumerate(many):
        assert torch.equal(merged['output_ids'][i, :b['output_ids'].shape[1]], b['output_ids'][0])
        assert (merged['pred_masks'][i] is None) == ('pred_masks' not in b)
        if 'pred_masks' in b:
            assert torch.equal(merged['pred_masks'][i], b['pred_masks'][0])                 # the same wave, the same kernels: bit-identical
    # waves: at most 3 rows decode together -> three waves, same results up to the kernels' choice for another row count
    waved = model.generate_avs_many(samples, max_rows=3, **kw)
    assert model._engine.last_plan["groups"] == [3, 3, 1]
    for i, (b, c) in enumerate(zip(many, waved)):
        if torch.equal(b['output_ids'], c['output_ids']) and 'pred_masks' in b:
            assert _rel(c['pred_masks'][0], b['pred_masks'][0], f"generate_avs_many in waves of 3, sample {i} (HIP vs HIP)") < 1.3e-2


def test_generate_avs_one_by_one_and_batched_vs_the_reference_loop_fixture():
    """tests/golden/avs_loop_tiny.npz = the REFERENCE's generate_avs looped over five one-sample calls (one clip under s4 / avss / ms3, two other
    clips without six mask tokens).  The product's generate_avs one sample at a time AND generate_avs_many over all five at once: ids equal to
    the reference's (a divergence only at a step whose reference... is not recorded here, so ids must simply be equal on the three margin-checked
    samples and may differ on the others only together with the mask outcome), the same samples produce masks, and the masks agree with the
    reference's within the SegModule's computed tolerance (tests/test_seg... rows: 2.1e-2 of scale vs the fp32 reference, as for seg_tiny.npz)."""
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    from tests.test_oracle_golden import _avs_loop_samples
    from tests.util import DuckTokenizer, bert_cfg
    meta, A = load_fixture("avs_loop_tiny")
    cfg = UnifiedConfig(**meta["dec"], pad_token_id=meta["pad_token_id"])
    cfg.vocab_size = meta["base_vocab"]
    model = get_peft_model(UnifiedForCausalLM(cfg, device="cuda"), LoraConfig())
    model.get_model().pad_token_id = meta["pad_token_id"]
    model.get_model().init_multimodal_modules(d_model=meta["d_model"], visual_branch=True, audio_branch=True, segment_branch=True, select_layer_list=meta["select"],
                                              clip_config=meta["clip"], beats_config=meta["beats"], bert_config=bert_cfg(meta["qf"]),
                                              vit_image_embedding_dim=meta["clip"]["hidden_size"])
    model.initialize_MM_tokenizer(DuckTokenizer(meta["base_vocab"]), mask_token_nums=6)
    r = model.load_state_dict(weights_from_table(meta), strict=False)
    assert not r.missing_keys, r.missing_keys[:5]
    model.SPECIAL_TOKEN_2_IDS.update({k: v for k, v in meta["special"].items() if k.startswith("<mask_")})
    n = meta["new_tokens"]
    kw = dict(max_new_tokens=n, pad_token_id=2, eos_token_id=None)
    samples = [dict(batch_input_ids=[ids], batch_labels=[torch.full_like(ids, -100)], batch_X_modals=[mods], batch_task_names=[task])
               for ids, mods, task in _avs_loop_samples(meta)]
    one = [model.generate_avs(**s_, **kw) for s_ in samples]
    many = model.generate_avs_many(samples, **kw)
    for form, res in (("one sample per call", one), ("generate_avs_many", many)):
        n_masks = 0
        for i, (r_, m) in enumerate(zip(res, meta["samples"])):
            same = torch.equal(r_["output_ids"].cpu(), A[f"ids_{i}"])
            if i < 3:
                assert same, (form, i, r_["output_ids"].tolist(), A[f"ids_{i}"].tolist())       # the clip the fixture's mask ids were taken from
            if not same:
                continue                                                                        # (an unsearched clip may flip a sub-margin step)
            assert ("pred_masks" in r_) == m["has_masks"], (form, i)
            if m["has_masks"]:
                pm = r_["pred_masks"][0]
                assert list(pm.shape) == m["shape"]
                sub = pm[:, 3::8, 5::8] if pm.shape[0] > 1 else pm[:, 1::2, ::2]
                assert _rel(sub, A[f"mask_sub_{i}"], f"avs_loop sample {i} ({m['task']}), {form}: masks vs the reference's generate_avs") < 2.1e-2
                n_masks += 1
        assert n_masks == 3, (form, n_masks)


def test_seg_module_batched_equals_sample_by_sample():
    """SegModule over a batch with mixed class counts (r06: samples of one class count run through the mask decoder together) on the reference
    fixture's two samples repeated so that BOTH contents appear in both class-count groups: every sample against the oracle on its own inputs
    (the tolerance of the one-sample test), against the same module called on that sample alone (HIP vs HIP: the GEMMs pick other tile shapes
    for other row counts and the mask decoder amplifies 1-ulp flips like any two executions do; nothing mixes rows of different samples - a
    mixed row would show as an O(1) error against the oracle), and against the fixture where the task matches."""
    from crab_amd.seg_module import SegModule
    from oracle import crab_oracle as O
    from tests.util import seg_inputs
    meta, A = load_fixture("seg_tiny")
    W = weights_from_table(meta)
    seg = SegModule(d_model=meta["d_model"], vit_image_embedding_dim=128, device="cuda")
    seg.load_state_dict({k[len("model.seg_module."):]: v for k, v in W.items()}, strict=True)
    pred, feats = seg_inputs(meta)
    order = [0, 1, 1, 0, 1, 0, 0]                                   # fixture samples, repeated and interleaved
    tasks = ['avss', 's4', 'avss', 'ms3', 'ref-avs', 'avss', 's4']  # avss group: contents 0, 1, 0; binary group: 1, 0, 1, 0
    P = pred[order].to(BF).cuda()
    F_ = [f[order].to(BF).cuda() for f in feats]
    batched = seg(pred_embeddings=P, multi_scale_image_feature_list=F_, low_res_mask_size=112, gt_mask=None, batch_task_names=tasks)['pred_masks']
    ref = O.seg_module(pred[order].to(BF).float(), [f[order].to(BF).float() for f in feats], tasks, _bf(W))
    for i, t in enumerate(tasks):
        solo = seg(pred_embeddings=P[i:i + 1], multi_scale_image_feature_list=[f[i:i + 1] for f in F_], low_res_mask_size=112, gt_mask=None,
                   batch_task_names=[t])['pred_masks'][0]
        assert batched[i].shape == solo.shape == ((71, 224, 224) if t == 'avss' else (1, 224, 224))
        assert _rel(batched[i], ref[i], f"SegModule batched sample {i} ({t}) vs oracle on bf16-rounded weights") < 1.3e-2
        assert _rel(batched[i], solo, f"SegModule batched sample {i} ({t}) vs the sample alone (HIP vs HIP)") < 1.3e-2
    assert _rel(batched[0][:, 3::8, 5::8], A["avss_sub"], "SegModule batched, avss sample vs fp32 reference") < 2.1e-2
    assert _rel(batched[1][:, 1::2, ::2], A["s4_sub"], "SegModule batched, s4 sample vs fp32 reference") < 2.1e-2
    seg.BATCH_AVSS, seg.BATCH_BINARY = 2, 2                         # chunked groups: every sample still its own result
    chunked = seg(pred_embeddings=P, multi_scale_image_feature_list=F_, low_res_mask_size=112, gt_mask=None, batch_task_names=tasks)['pred_masks']
    for i in range(len(tasks)):
        assert _rel(chunked[i], ref[i], f"SegModule in chunks of 2, sample {i} vs oracle") < 1.3e-2


def test_harness_pixel_task_loop_writes_the_masks(tmp_path):
    """harness.run_inference_avs = scripts/quick_start.py:270-450: generate_avs -> text + mask files.  Binary task: a mode-'P' PNG with 255 where
    sigmoid(pred) > 0.5; avss: an RGB PNG of palette[argmax over the 71 class planes]; a sample without the six mask tokens: no file."""
    import json
    import numpy as np
    from PIL import Image
    from crab_amd import harness, ops
    from tests.util import DuckTokenizer
    model, meta, A, W, sp, ids, image, mods, lab, n, plain = _avs_setup()

    class Tok(DuckTokenizer):
        def decode(self, ids_, skip_special_tokens=False):
            return " ".join(str(int(i)) for i in ids_)
    tok = Tok(meta["base_vocab"])
    mk = lambda task, path: {"batch_input_ids": [ids], "batch_labels": lab, "batch_X_modals": mods, "batch_task_names": [task],
                             "batch_metadata": [{"instruction": "seg", "output": "x", "image_path": "/d/v1/frames/3.jpg", "mask_path": path}]}
    batches = [mk("s4", "/data/avs/vidA/0/3.png"), mk("avss", "/data/avs/vidB/0/7.png")]
    recs = harness.run_inference_avs(batches, model, tok, str(tmp_path), max_new_tokens=n, pad_token_id=2, eos_token_id=None,
                                     out_path=str(tmp_path / "res.jsonl"))
    assert [r["num_classes"] for r in recs] == [1, 71] and recs[0]["predict"] == " ".join(str(int(i)) for i in plain[0])
    direct = [model.generate_avs(batch_input_ids=[ids], batch_labels=lab, batch_X_modals=mods, batch_task_names=[t], max_new_tokens=n, pad_token_id=2,
                                 eos_token_id=None)["pred_masks"][0].float() for t in ("s4", "avss")]
    p0 = np.array(Image.open(recs[0]["pred_path"]))
    assert recs[0]["pred_path"].endswith("mask_img_dir/vidA/3_pred.png") and Image.open(recs[0]["pred_path"]).mode == "P"
    assert np.array_equal(p0, ((torch.sigmoid(direct[0][0]) > 0.5).cpu().numpy() * 255).astype(np.uint8))
    rnd = torch.randn(1, 224, 224, device="cuda")                              # (the tiny random model's mask is one-signed: both signs here)
    assert np.array_equal(ops.mask_labels(rnd).cpu().numpy(), ((rnd[0] > 0).cpu().numpy() * 255).astype(np.uint8))
    p1 = np.array(Image.open(recs[1]["pred_path"]))
    cls = torch.argmax(torch.softmax(direct[1], 0), 0).cpu().numpy()
    assert recs[1]["pred_path"].endswith("avss_result/vidB/7_pred.png") and np.array_equal(p1, harness.default_palette()[cls]) and len(np.unique(cls)) > 1
    assert np.array_equal(ops.mask_labels(direct[1]).cpu().numpy(), cls.astype(np.uint8))
    assert len(open(tmp_path / "res.jsonl").read().strip().splitlines()) == 2
    assert "iou" not in recs[0] and "_avss" not in recs[1]                     # no ground truth in X_modals: no metrics
    # with the ground truth '<mask>' in X_modals the loop scores each sample on the device (utils/avss_utils.py through crab_amd.avss_utils):
    # the records and the closing averages equal the CPU restatement on the same masks
    from oracle import metrics_oracle as MO
    g = torch.Generator().manual_seed(3)
    gt_bin = (torch.rand(1, 224, 224, generator=g) > 0.5).float()
    gt_cls = torch.randint(0, 71, (1, 224, 224), generator=g)
    gt_cls[0, :100] = torch.argmax(direct[1], 0)[:100].cpu()
    withgt = lambda task, path, m: dict(mk(task, path), batch_X_modals=[dict(mods[0], **{"<mask>": m})])
    summ = {}
    rec2 = harness.run_inference_avs([withgt("s4", "/data/avs/vidD/0/3.png", gt_bin), withgt("ms3", "/data/avs/vidD/0/4.png", 1 - gt_bin),
                                      withgt("avss", "/data/avs/vidE/0/7.png", gt_cls)], model, tok, str(tmp_path), max_new_tokens=n, pad_token_id=2,
                                     eos_token_id=None, summary=summ, out_path=str(tmp_path / "res2.jsonl"))
    d0 = direct[0].cpu().numpy()
    for r, gtm in ((rec2[0], gt_bin), (rec2[1], 1 - gt_bin)):
        assert r["iou"] == float(MO.mask_iou(d0, gtm.numpy())) and r["fscore"] == MO.eval_fmeasure(d0, gtm.numpy())
    assert summ["count"] == 2 and summ["miou"] == float(np.float32(np.float32(rec2[0]["iou"]) + np.float32(rec2[1]["iou"])) / np.float32(2))
    assert abs(summ["f_score"] - (rec2[0]["fscore"] + rec2[1]["fscore"]) / 2) < 1e-12
    want = MO.avss_final(*MO.batch_miou_fscore(direct[1].cpu().numpy()[None], gt_cls.numpy())[:3])
    assert summ["avss"]["count"] == 1 and all(abs(summ["avss"][k] - want[k]) <= 1e-6 for k in want) and want["miou"] > 0
    assert "_avss" not in rec2[2] and all("_avss" not in json.loads(l) for l in open(tmp_path / "res2.jsonl"))
    # the ground truths beside the predictions (quick_start.py:104-109, avss_utils.py save_gt_mask)
    assert rec2[0]["gt_path"].endswith("mask_img_dir/vidD/3_gt.png") and Image.open(rec2[0]["gt_path"]).mode == "P"
    assert np.array_equal(np.array(Image.open(rec2[0]["gt_path"])), (gt_bin[0].numpy() * 255).astype(np.uint8))
    assert np.array_equal(np.array(Image.open(rec2[2]["gt_path"])), harness.default_palette()[gt_cls[0].numpy()]) and "gt_path" not in recs[0]
    summ_n = {}
    rn = harness.run_inference_avs([withgt("ref-avs", "/data/avs/vidF/0/1.png", torch.zeros(1, 224, 224))], model, tok, str(tmp_path), max_new_tokens=n,
                                   pad_token_id=2, eos_token_id=None, null_reference=True, summary=summ_n)
    assert rn[0]["s"] == float(MO.metric_s_for_null(d0)) == summ_n["ms"] and "iou" not in rn[0]
    # coalesce = True (r06): the same three samples as ONE generate_avs_many call - same records, metrics within what the mask decoder's other
    # tile shapes leave (the thresholded maps may differ in pixels whose logit is within bf16 noise of 0)
    summ_c = {}
    rc = harness.run_inference_avs([withgt("s4", "/data/avs/vidG/0/3.png", gt_bin), withgt("ms3", "/data/avs/vidG/0/4.png", 1 - gt_bin),
                                    withgt("avss", "/data/avs/vidH/0/7.png", gt_cls)], model, tok, str(tmp_path), max_new_tokens=n, pad_token_id=2,
                                   eos_token_id=None, summary=summ_c, coalesce=True, coalesce_rows=8)
    assert [r["predict"] for r in rc] == [r["predict"] for r in rec2] and [r["num_classes"] for r in rc] == [1, 1, 71]
    assert all(abs(a["iou"] - b["iou"]) < 2e-3 and abs(a["fscore"] - b["fscore"]) < 2e-3 for a, b in zip(rc[:2], rec2[:2]))
    assert summ_c["count"] == 2 and abs(summ_c["miou"] - summ["miou"]) < 2e-3 and abs(summ_c["avss"]["miou"] - summ["avss"]["miou"]) < 2e-3
    assert rc[2]["pred_path"].endswith("avss_result/vidH/7_pred.png") and np.mean(np.array(Image.open(rc[2]["pred_path"])) != np.array(Image.open(rec2[2]["pred_path"]))) < 3e-2    # (71-way argmax of a random model: near-ties flip; measured 1.0e-2)
    nop = harness.run_inference_avs([withgt("s4", "/data/avs/vidI/0/3.png", gt_bin)] * 2, model, tok, str(tmp_path), max_new_tokens=n, pad_token_id=2,
                                    eos_token_id=None, coalesce=True, write_png=False)
    assert nop[0]["pred_path"] is None and "gt_path" not in nop[0] and abs(nop[1]["iou"] - rec2[0]["iou"]) < 2e-3 and not (tmp_path / "mask_img_dir" / "vidI").exists()
    # no mask tokens in the output -> no masks, no file, the record says so (quick_start.py:303-306)
    for i in range(6):
        sp[f'<mask_{i}>'] = meta["base_vocab"] + 11 + i
    r = harness.run_inference_avs([mk("s4", "/data/avs/vidC/0/1.png")], model, tok, str(tmp_path), max_new_tokens=n, pad_token_id=2, eos_token_id=None)
    assert r[0]["pred_path"] is None and not (tmp_path / "mask_img_dir" / "vidC").exists()


def test_encoder_chunking_does_not_change_the_features():
    """prepare_multimodal_inputs encodes the modality blocks in chunks of ENC_CHUNK clips (bounded scratch at hundreds of clips per call): the
    rows are independent, so the spliced inputs_embeds of 5 clips agree for chunks of 2 and for one call over all of them - to the last bit wherever
    both runs take the same kernels (every realistic chunk is deep inside the large-M regime; at this tiny size the 2-clip chunk crosses a tile-choice
    boundary of the GEMM dispatcher, which changes the summation order: agreement to bf16 accumulation noise is what is asserted)."""
    from crab_amd import synth, unified_arch
    meta, A = load_fixture("full_tiny_llama")
    model = build_tiny_crab(meta)
    model.load_state_dict(weights_from_table(meta), strict=False)
    um = model.base_model.model
    p = meta["prompts"]
    ids = [synth.synth_prompt_ids(20 + i, meta["base_vocab"], um.SPECIAL_TOKEN_2_IDS, seed=meta["seed"], clip=50 + i) for i in range(5)]
    mods = [{'<video>': synth.synth_video(p["t_v"], seed=meta["seed"], clip=50 + i), '<audio>': synth.synth_audio(p["t_a"], p["l_a"], seed=meta["seed"], clip=50 + i)}
            for i in range(5)]
    lab = [torch.full_like(i, -100) for i in ids]
    outs = []
    saved = unified_arch.ENC_CHUNK
    try:
        for chunk in (64, 2):
            unified_arch.ENC_CHUNK = chunk
            outs.append(um.prepare_multimodal_inputs(ids, lab, mods, ['avqa'] * 5)["inputs_embeds"].clone())
    finally:
        unified_arch.ENC_CHUNK = saved
    # one bf16 ulp of the largest embedding (|x| in [2, 4): 2^-6) is 5.3e-3 of the scale and is what a changed summation order produces: two ulps allowed
    assert _rel(outs[1], outs[0].float().cpu(), "inputs_embeds: encoder chunks of 2 clips vs one call over 5 (HIP vs HIP)") < 1.1e-2


def test_generate_many_clips_vs_oracle():
    """Six more synthetic clips (3 batches of 2, different prompts / frames / fbank) through the public generate() against
    the golden-pinned oracle run on the same bf16-rounded weights: greedy ids exact wherever the oracle's top-2 margin
    exceeds twice the measured logit error."""
    from crab_amd import synth
    from oracle import crab_oracle as O
    meta, A = load_fixture("full_tiny_llama")
    W = weights_from_table(meta)
    model = build_tiny_crab(meta)
    model.load_state_dict(W, strict=False)
    Wo = _bf(O.strip_peft_prefix(W))
    from tests.test_oracle_golden import _full_cfg
    ocfg = _full_cfg(meta)
    p = meta["prompts"]
    n = 12
    worst = 0.0
    for c0 in (11, 23, 37):
        ids = [synth.synth_prompt_ids(nt, ocfg.base_vocab, model.SPECIAL_TOKEN_2_IDS, seed=meta["seed"], clip=c)
               for nt, c in ((p["n0"], c0), (p["n1"], c0 + 1))]                       # ragged: left padding inside the batch
        mods = [{'<video>': synth.synth_video(p["t_v"], seed=meta["seed"], clip=c), '<audio>': synth.synth_audio(p["t_a"], p["l_a"], seed=meta["seed"], clip=c)}
                for c in (c0, c0 + 1)]
        lab = [torch.full_like(i, -100) for i in ids]
        r = model.generate(batch_input_ids=ids, batch_labels=lab, batch_X_modals=mods, batch_task_names=['avqa'] * 2, use_cache=True,
                           max_new_tokens=n, pad_token_id=2, eos_token_id=None, output_logits=True, return_dict_in_generate=True)
        ref_ids, ref_logits = O.generate(ids, mods, Wo, ocfg, n)
        # these clips are NOT searched for wide margins (unlike the fixtures): with the fp32 residual stream (r04) clip 11 flips one step whose
        # reference top-2 margin is below twice the logit error (asserted inside) - 20 of 21 steps covered there, all steps elsewhere
        worst = max(worst, _check_ids(r.sequences, ref_ids, ref_logits, torch.stack(r.logits, 1), min_frac=0.9) / ref_logits.abs().max().item())
    assert worst < PB.bound("full_tiny_llama: end to end"), worst          # (other clips of the fixture's stack: its end-to-end row)


def test_sharp_margin_clips_decode_to_the_reference_ids_without_escape():
    """north_star's ids clause where it is decidable: tests/golden/sharp_tiny_llama.npz = eight clips recorded from the reference whose top-2
    logit margin is >= 0.15 on EVERY greedy step (10 x the bf16 logit error of this stack; 8 of the first 154 candidate clips qualify).  Greedy
    ids must be EQUAL to the reference's on every step of every clip - torch.equal, no margin escape - alone (bs 1), all eight in one generate()
    (left-padded batch: the reference's generate attends the pads, so only the ids of the longest prompt are comparable there), and as eight
    calls coalesced into one ragged decode batch; the per-step logits stay inside the computed end-to-end bound."""
    from crab_amd import synth
    meta, A = load_fixture("sharp_tiny_llama")
    model = build_tiny_crab(meta)
    model.load_state_dict(weights_from_table(meta), strict=False)
    n = meta["new_tokens"]
    scale = A["logits"].abs().max().item()
    batches = []
    worst = 0.0
    for i, (c, nt) in enumerate(zip(meta["clips"], meta["prompt_tokens"])):
        ids = synth.synth_prompt_ids(nt, meta["base_vocab"], meta["special"], seed=meta["seed"], clip=c)
        mods = [{'<video>': synth.synth_video(meta["t_v"], seed=meta["seed"], clip=c), '<audio>': synth.synth_audio(meta["t_a"], meta["l_a"], seed=meta["seed"], clip=c)}]
        b = dict(batch_input_ids=[ids], batch_labels=[torch.full_like(ids, -100)], batch_X_modals=mods, batch_task_names=['avqa'])
        batches.append(b)
        r = model.generate(**b, use_cache=True, max_new_tokens=n, pad_token_id=2, eos_token_id=None, output_logits=True, return_dict_in_generate=True)
        assert torch.equal(r.sequences[0].cpu(), A["ids"][i]), (c, r.sequences[0].tolist(), A["ids"][i].tolist())
        worst = max(worst, (torch.stack(r.logits, 1)[0].float().cpu() - A["logits"][i]).abs().max().item())
    from tests.util import record_parity
    record_parity("sharp-margin clips (8 clips x 8 steps, reference margins >= 0.15): ids EQUAL on every step; worst per-step logit error", worst, scale,
                  PB.bound("full_tiny_llama: end to end"), min_ref_margin=float(A["margin"].min()), margin_over_error=float(A["margin"].min()) / worst)
    assert worst < PB.bound("full_tiny_llama: end to end") * scale and float(A["margin"].min()) > 5 * worst
    many = model.generate_batches(batches, coalesce=True, use_cache=True, max_new_tokens=n, pad_token_id=2, eos_token_id=None)
    for i, ids in enumerate(many):
        assert torch.equal(ids[0].cpu(), A["ids"][i]), ("coalesced", meta["clips"][i])


def test_unfiltered_id_parity_statistic_on_unsearched_clips():
    """A denominator nobody selected: the 24 clips of tests/golden/id_stats_tiny_llama.npz (indices 200..223, recorded from the
    reference's generate() without any margin search), one clip per generate() call.  REPORTED in the parity report: the fraction of greedy
    steps identical to the reference's, how many clips agree on all 12 steps, and at each first divergence the reference's top-2 margin next to
    the logit error there.  Gated only on what must hold physically: a divergence happens where margin <= 2 x the logit error, and the logit error
    stays under the computed end-to-end bound of this stack (tests/bounds.py)."""
    from crab_amd import synth
    from tests.util import record_parity
    meta, A = load_fixture("id_stats_tiny_llama")
    model = build_tiny_crab(meta)
    model.load_state_dict(weights_from_table(meta), strict=False)
    n = meta["new_tokens"]
    scale = A["logits"].abs().max().item()
    steps_same = clips_same = 0
    divergences, worst = [], 0.0
    for i, (c, nt) in enumerate(zip(meta["clips"], meta["prompt_tokens"])):
        ids = synth.synth_prompt_ids(nt, meta["base_vocab"], meta["special"], seed=meta["seed"], clip=c)
        mods = [{'<video>': synth.synth_video(meta["t_v"], seed=meta["seed"], clip=c), '<audio>': synth.synth_audio(meta["t_a"], meta["l_a"], seed=meta["seed"], clip=c)}]
        r = model.generate(batch_input_ids=[ids], batch_labels=[torch.full_like(ids, -100)], batch_X_modals=mods, batch_task_names=['avqa'], use_cache=True,
                           max_new_tokens=n, pad_token_id=2, eos_token_id=None, output_logits=True, return_dict_in_generate=True)
        got_ids, got_logits = r.sequences[0].cpu(), torch.stack(r.logits, 1)[0].float().cpu()
        k = n
        for s in range(n):
            e = (got_logits[s] - A["logits"][i, s]).abs().max().item()      # contexts identical up to the first divergence
            worst = max(worst, e)
            if got_ids[s] != A["ids"][i, s]:
                m_ = A["margin"][i, s].item()
                divergences.append({"clip": c, "step": s, "ref_margin": round(m_, 5), "logit_err": round(e, 5)})
                assert m_ <= 2 * e, (c, s, m_, e)
                k = s
                break
        steps_same += k
        clips_same += k == n
    total = n * len(meta["clips"])
    record_parity("UNSEARCHED clips 200..223 (tiny Llama, bs 1): greedy steps identical to the reference's before the first divergence / all steps",
                  worst, scale, None, steps_identical=steps_same, steps_total=total, fraction=round(steps_same / total, 4), clips_fully_identical=clips_same,
                  clips=len(meta["clips"]), first_divergences=divergences, min_ref_margin=float(A["margin"].min()), median_ref_margin=float(A["margin"].median()))
    assert worst < PB.bound("full_tiny_llama: end to end") * scale, worst
    assert steps_same >= 0.5 * total, (steps_same, total, divergences)            # sanity only: the statistic itself is the report row


def test_generate_edge_cases_vs_oracle():
    """Prompts the AVQA fixture does not cover: text only, video only, audio only, an <image> block, one- and two-token
    generations (no HIP graph below three tokens), EOS on the very first token - each against the golden-pinned oracle."""
    from crab_amd import synth
    from oracle import crab_oracle as O
    from tests.test_oracle_golden import _full_cfg
    meta, A = load_fixture("full_tiny_llama")
    W = weights_from_table(meta)
    model = build_tiny_crab(meta)
    model.load_state_dict(W, strict=False)
    Wo = _bf(O.strip_peft_prefix(W))
    ocfg = _full_cfg(meta)
    sp = model.SPECIAL_TOKEN_2_IDS
    p = meta["prompts"]
    g = torch.Generator().manual_seed(123)
    text = torch.randint(3, ocfg.base_vocab, (19,), generator=g)
    vid = synth.synth_video(p["t_v"], seed=meta["seed"], clip=41)
    aud = synth.synth_audio(p["t_a"], p["l_a"], seed=meta["seed"], clip=41)
    def with_block(kind):
        ids = text.clone()
        ids[4:7] = torch.tensor([sp[f"<{kind}_start>"], sp[f"<{kind}>"], sp[f"<{kind}_end>"]])
        return ids
    cases = [("text only", text.clone(), {}),
             ("video only", with_block("video"), {'<video>': vid}),
             ("audio only", with_block("audio"), {'<audio>': aud}),
             ("image", with_block("image"), {'<image>': vid[:1]})]
    for name, ids, mod in cases:
        for n in (1, 2, 5):
            r = model.generate(batch_input_ids=[ids], batch_labels=[torch.full_like(ids, -100)], batch_X_modals=[mod], batch_task_names=['avqa'],
                               use_cache=True, max_new_tokens=n, pad_token_id=2, eos_token_id=None, output_logits=True, return_dict_in_generate=True)
            ref_ids, ref_logits = O.generate([ids], [mod], Wo, ocfg, n)
            assert r.sequences.shape == (1, n), name
            _check_ids(r.sequences, ref_ids, ref_logits, torch.stack(r.logits, 1))
    # EOS == the first generated token: one column, generation stops immediately (HF semantics)
    ids, mod = cases[1][1], cases[1][2]
    first = int(O.generate([ids], [mod], Wo, ocfg, 1)[0][0, 0])
    out = model.generate(batch_input_ids=[ids], batch_labels=[torch.full_like(ids, -100)], batch_X_modals=[mod], batch_task_names=['avqa'],
                         max_new_tokens=20, pad_token_id=2, eos_token_id=first)
    assert out.shape == (1, 1) and int(out[0, 0]) == first


def test_llama_ops_fixture_rmsnorm_rope_layer_prefill():
    """tests/golden/llama_ops.npz (the reference's in-tree modeling_llama.py): RMSNorm and RoPE kernels, and one hyper-LoRA
    decoder layer run as prefill through the engine (residual stream before the final norm, K / V cache rows)."""
    from crab_amd import ops
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    meta, A = load_fixture("llama_ops")
    c = meta["cfg"]
    # RMSNorm (bf16 in / out against the fp32 reference)
    x = A["norm_x"].reshape(-1, c["hidden_size"]).to(BF).cuda()
    y = ops.rmsnorm(x, A["norm_w"].to(BF).cuda(), c["rms_norm_eps"])
    assert _rel(y, A["norm_y"].reshape(-1, c["hidden_size"]), "llama_ops: rmsnorm") < 1.2e-2
    # RoPE: rows at positions 0..3 as one sequence, then the rows at 9, 17, 40 one by one (pos0 = absolute position)
    H, d = c["num_attention_heads"], 64
    tab = ops.rope_table(64, d, c["rope_theta"], "cuda")
    q, k, pos = A["rope_q"], A["rope_k"], A["rope_pos"][0].tolist()

    def run(rows, pos0):
        S = len(rows)
        qkv = torch.cat([q[0, :, rows].transpose(0, 1).reshape(S, H * d), k[0, :, rows].transpose(0, 1).reshape(S, H * d),
                         torch.zeros(S, H * d)], dim=1).to(BF).cuda().contiguous()
        kc = torch.zeros(1, H, 64, d, device="cuda", dtype=BF)
        vc = torch.zeros_like(kc)
        ops.qkv_rope_split(qkv, tab, kc, vc, None, 1, S, H, H, d, 64, pos0=pos0)
        got_q = qkv[:, :H * d].reshape(S, H, d).transpose(0, 1)
        got_k = kc[0, :, pos0:pos0 + S]
        assert _rel(got_q, A["rope_q_out"][0][:, rows], "llama_ops: rope q") < 1.2e-2
        assert _rel(got_k, A["rope_k_out"][0][:, rows], "llama_ops: rope k") < 1.2e-2

    run([0, 1, 2, 3], 0)
    for r in (4, 5, 6):
        run([r], pos[r])
    _layer_prefill_and_decode_step("llama_ops", False)


def _layer_prefill_and_decode_step(fixture, qwen):
    """One hyper-LoRA decoder layer of the reference's vendored modeling file, run through the engine as a prefill of S rows and then
    as a 1-token decode step against the cache the prefill filled (layer output = residual stream before the final norm; K / V rows)."""
    from crab_amd import ops
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    meta, A = load_fixture(fixture)
    c = dict(meta["cfg"])
    if qwen:
        from crab_amd.unified_qwen import UnifiedConfig, UnifiedForCausalLM
        c.update(attention_bias=True)
    else:
        from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    c.update(num_hidden_layers=1, vocab_size=320, pad_token_id=2)
    model = get_peft_model(UnifiedForCausalLM(UnifiedConfig(**c), device="cuda"), LoraConfig())
    W = {"base_model.model." + k_: v for k_, v in weights_from_table(meta).items()}
    r = model.load_state_dict(W, strict=False)
    assert not r.unexpected_keys, r.unexpected_keys[:5]
    um = model.base_model.model
    eng = um._engine
    S, D = A["layer_x"].shape[1], A["layer_x"].shape[2]
    kc, vc = eng.alloc_cache(1, 64)
    eng.prefill(A["layer_x"].to(BF).cuda(), kc, vc, b0=0, all_logits=True)      # every row through the layer (generate()'s prefill finishes the last rows only)
    ws = eng._workspace(S)
    assert _rel(ws.x[:S], A["layer_y"][0], f"{fixture}: layer output, prefill") < PB.bound(f"{fixture}: one hyper-LoRA decoder layer")
    assert _rel(kc[0, 0, :, :S], A["cache_k"][0][:, :S], f"{fixture}: K cache rows, prefill") < 1.2e-2
    assert _rel(vc[0, 0, :, :S], A["cache_v"][0][:, :S], f"{fixture}: V cache rows, prefill") < 1.2e-2
    # the decode row: position S, device-resident position word, KV append fused behind the q|k|v projection
    ops.cast_rows(A["layer_x1"][0].to(BF).cuda(), ws.x, 1, D)
    pos = torch.full((1,), S, device="cuda", dtype=torch.int32)
    x, _ = eng._layers(ws, 1, 1, kc, vc, 0, 64, 0, pos, None)
    assert _rel(x[:1], A["layer_y1"][0], f"{fixture}: layer output, 1-token decode step") < PB.bound(f"{fixture}: one hyper-LoRA decoder layer")
    assert _rel(kc[0, 0, :, S], A["cache_k"][0][:, S], f"{fixture}: K cache row, decode step") < 1.2e-2
    assert _rel(vc[0, 0, :, S], A["cache_v"][0][:, S], f"{fixture}: V cache row, decode step") < 1.2e-2


def test_qwen_ops_fixture_rmsnorm_rope_layer_prefill_and_decode():
    """tests/golden/qwen_ops.npz (the reference's vendored models/qwen/modeling_qwen2.py): RMSNorm at eps 1e-6, RoPE at theta 1e6 up to
    position 1400, one hyper-LoRA Qwen2 layer (GQA 4/2, q/k/v bias) as prefill + decode step."""
    from crab_amd import ops
    meta, A = load_fixture("qwen_ops")
    c = meta["cfg"]
    D, H, Hk = c["hidden_size"], c["num_attention_heads"], c["num_key_value_heads"]
    d = D // H
    x = A["norm_x"].reshape(-1, D).to(BF).cuda()
    y = ops.rmsnorm(x, A["norm_w"].to(BF).cuda(), c["rms_norm_eps"])
    assert _rel(y, A["norm_y"].reshape(-1, D), "qwen_ops: rmsnorm") < 1.2e-2
    tab = ops.rope_table(1408, d, c["rope_theta"], "cuda")
    q, k, pos = A["rope_q"], A["rope_k"], A["rope_pos"][0].tolist()
    for r_ in range(len(pos)):
        qkv = torch.cat([q[0, :, r_].reshape(1, H * d), k[0, :, r_].reshape(1, Hk * d), torch.zeros(1, Hk * d)], dim=1).to(BF).cuda().contiguous()
        kc = torch.zeros(1, Hk, 1408, d, device="cuda", dtype=BF)
        vc = torch.zeros_like(kc)
        ops.qkv_rope_split(qkv, tab, kc, vc, None, 1, 1, H, Hk, d, 1408, pos0=pos[r_])
        assert _rel(qkv[:, :H * d].reshape(H, d), A["rope_q_out"][0][:, r_], f"qwen_ops: rope q @ {pos[r_]}") < 1.2e-2
        assert _rel(kc[0, :, pos[r_]], A["rope_k_out"][0][:, r_], f"qwen_ops: rope k @ {pos[r_]}") < 1.2e-2
    _layer_prefill_and_decode_step("qwen_ops", True)


def _gen_both_sequencers(model, embeds, n):
    """generate() with the layers sequenced by crab_llama_layers (C) and by crab_amd/decoder.py (Python): same launches, same order."""
    from crab_amd import decoder, ops
    out = []
    calls = [0]
    real = ops.llama_layers

    def counted(*a):
        calls[0] += 1
        return real(*a)
    for native in (True, False):
        decoder.NATIVE_LAYERS = native
        ops.llama_layers = counted
        calls[0] = 0
        try:
            for use_graph in (True, False):
                eng = model.base_model.model._engine
                eng._dec.clear()                                  # a decode state keeps its captured graph: capture again
                r = eng.generate(embeds, n, eos_token_id=None, pad_token_id=2, return_step_logits=True, use_graph=use_graph)
                out.append((r[0].clone(), r[1].clone()))
        finally:
            decoder.NATIVE_LAYERS = True
            ops.llama_layers = real
        assert (calls[0] > 0) == native, (native, calls[0])      # the C sequencer really is the one that ran (or did not)
    return out


@pytest.mark.parametrize("bs", [1, 3])
def test_native_layer_sequencer_equals_python_sequence_tiny_qwen(bs):
    """csrc/llama_layer.hip (crab_llama_layers) against the per-launch Python sequence on the tiny Qwen2 decoder (GQA, q|k|v bias,
    hyper-LoRA on every projection): bit-identical ids and per-step logits, graph replay and eager."""
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_qwen import UnifiedConfig, UnifiedForCausalLM
    meta, A = load_fixture("decoder_tiny_qwen2")
    W = weights_from_table(meta)
    cfg = UnifiedConfig(**meta["dec"], attention_bias=True, pad_token_id=2)
    model = get_peft_model(UnifiedForCausalLM(cfg, device="cuda"), LoraConfig())
    model.load_state_dict(W, strict=False)
    emb = A["embeds"].to(BF).cuda()
    emb = torch.cat([emb] + [emb.flip(1) * (0.5 + 0.25 * i) for i in range(bs - 1)], 0) if bs > 1 else emb
    outs = _gen_both_sequencers(model, emb, 5)
    for ids, logits in outs[1:]:
        assert torch.equal(ids, outs[0][0]) and torch.equal(logits, outs[0][1])


@pytest.mark.parametrize("r,nl", [(16, 3), (4, 8)])
def test_adapter_rank_outside_the_small_batch_tail_limits_still_generates(r, nl):
    """lora_r is a CLI argument of the reference (peft_hyper/tuners/lora.py:42-83; 8 is only the default).  r = 16 (nl + r > 16) and nl * r = 32 at
    the edge: batches <= 16 must take the router + K-extension path when crab_rowfin_lora_ok says no (r03 hard-failed with CRAB_E_UNSUPPORTED),
    in BOTH sequencers (csrc/llama_layer.hip run_group and crab_amd/peft_hyper.py), and agree with the oracle at batch 1, 3 and 20."""
    from crab_amd import ops
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    from oracle import crab_oracle as O
    torch.manual_seed(11)
    cfg = UnifiedConfig(hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2,
                        vocab_size=320, pad_token_id=2)
    um = UnifiedForCausalLM(cfg, device="cuda")
    model = get_peft_model(um, LoraConfig(r=r, lora_alpha=2 * r, lora_nums=nl))
    assert bool(ops.rowfin_lora_ok(nl, r, 128)) == (nl + r <= 16 and nl * r <= 32)
    for n_, p in model.named_parameters():
        small = 0.2 if ("o_proj" in n_ or "down_proj" in n_ or "lora_B" in n_) else 1.0
        p.data.copy_((torch.randn(p.shape) * 0.08 * small).to(BF) if p.dim() > 1 else (1 + 0.1 * torch.randn(p.shape)).to(BF))
    W = {k: v.detach().float().cpu() for k, v in O.strip_peft_prefix(model.state_dict()).items() if v.dtype.is_floating_point}
    ocfg = O.DecoderConfig(hidden_size=128, intermediate_size=352, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, vocab_size=320,
                           lora_r=r, lora_alpha=2 * r, lora_nums=nl)
    for B in (1, 3, 20):
        emb = (torch.randn(B, 7, 128) * 0.5).to(BF).cuda()
        outs = _gen_both_sequencers(model, emb, 4)
        for ids, logits in outs[1:]:
            assert torch.equal(ids, outs[0][0]) and torch.equal(logits, outs[0][1])
        ref_ids, ref_logits = O.greedy_generate(emb.float().cpu(), W, ocfg, 4)
        err = _check_ids(outs[0][0], ref_ids, ref_logits, outs[0][1], min_frac=0.5)
        assert err < PB.decoder_bound(emb.cpu(), W, ocfg, ref_ids) * ref_logits.abs().max().item(), (B, err)


def test_decode_batch_between_256_and_512_rows_two_row_groups_tiny():
    """A decode batch of 300 rows (two row groups, the second one ragged: 44 rows) on the tiny hyper-LoRA Llama through generate()'s captured graph:
    the C sequencer and the Python sequence agree bit for bit, rows are independent of the batch they decode in (rows 0, 255, 256, 299 vs the same
    prompts decoded as a batch of 4: the skinny kernels - same ids wherever the margin allows, logits within the decoder tolerance), and the oracle
    agrees on those rows."""
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    from oracle import crab_oracle as O
    meta, A = load_fixture("full_tiny_llama")
    W = weights_from_table(meta)
    cfg = UnifiedConfig(**meta["dec"], pad_token_id=meta["pad_token_id"])
    model = get_peft_model(UnifiedForCausalLM(cfg, device="cuda"), LoraConfig())
    model.load_state_dict({k: v for k, v in W.items() if ".layers." in k or "embed_tokens" in k or "lm_head" in k or k.endswith("model.norm.weight")}, strict=False)
    g = torch.Generator().manual_seed(17)
    B, S, n = 300, 9, 5
    emb = (torch.randn(B, S, cfg.hidden_size, generator=g) * 0.5).to(BF).cuda()
    outs = _gen_both_sequencers(model, emb, n)
    for ids, logits in outs[1:]:
        assert torch.equal(ids, outs[0][0]) and torch.equal(logits, outs[0][1])
    ids, logits = outs[0][0].cpu(), outs[0][1].float().cpu()
    rows = [0, 255, 256, 299]
    eng = model.base_model.model._engine
    sids, slog = eng.generate(emb[rows], n, eos_token_id=None, pad_token_id=2, return_step_logits=True)
    Wo = {k: v for k, v in _bf(O.strip_peft_prefix(W)).items() if v.dtype.is_floating_point}
    ocfg = O.DecoderConfig(**meta["dec"])
    ref_ids, ref_logits = O.greedy_generate(emb[rows].float().cpu(), Wo, ocfg, n)
    worst = _check_ids(ids[rows], ref_ids, ref_logits, logits[rows], min_frac=0.75)
    bnd = PB.decoder_bound(emb[rows].cpu(), Wo, ocfg, ref_ids, W_stored=Wo)          # (the oracle above runs on the stored parameters: so does the bound)
    assert worst < bnd * ref_logits.abs().max().item(), worst
    _check_ids(sids.cpu(), ref_ids, ref_logits, slog.float().cpu(), min_frac=0.75)
    assert _rel(logits[rows][:, 0], slog.float().cpu()[:, 0], "tiny Llama: rows of a 300-row decode batch vs the same rows as a batch of 4, first step (HIP vs HIP)") < 2 * bnd


def test_single_layer_entry_points_equal_the_stack_call():
    """crab_llama_layer_prefill / crab_llama_layer_decode called layer by layer == crab_llama_layers over the table (tiny Llama,
    hyper-LoRA): x and h after the stack bit-identical, prefill (S = 9) and one decode step; argument validation of the io block."""
    import ctypes as C
    from crab_amd import _lib, ops
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    torch.manual_seed(5)
    cfg = UnifiedConfig(hidden_size=128, intermediate_size=352, num_hidden_layers=3, num_attention_heads=2, num_key_value_heads=2,
                        vocab_size=320, pad_token_id=2)
    um = UnifiedForCausalLM(cfg, device="cuda")
    model = get_peft_model(um, LoraConfig())
    for p in model.parameters():
        p.data.copy_((torch.randn(p.shape) * (0.05 if p.dim() > 1 else 1.0)).to(BF) if p.dim() > 1 else (1 + 0.1 * torch.randn(p.shape)).to(BF))
    eng = um._engine
    B, S, D = 2, 9, cfg.hidden_size
    emb = (torch.randn(B, S, D) * 0.5).to(BF).cuda()
    lib, dev = _lib.load(), eng.device
    stream = torch.cuda.current_stream().cuda_stream

    def run(per_layer: bool, decode: bool, kc, vc):
        ws = eng._workspace(B * (1 if decode else S))
        M = B * (1 if decode else S)
        ops.cast_rows((emb[:, -1] if decode else emb).reshape(M, D).contiguous(), ws.x, M, D)
        ops.rmsnorm(ws.x[:M], eng.model.layers[0].input_layernorm.weight, cfg.rms_norm_eps, out=ws.h[:M])
        Tmax = kc.shape[3]
        vt = None if decode else torch.zeros((B, cfg.num_key_value_heads, cfg.hidden_size // cfg.num_attention_heads, 16), device="cuda", dtype=BF)
        pos = torch.full((1,), S, device="cuda", dtype=torch.int32) if decode else None
        if not per_layer:
            eng._layers_native(ws, B, 1 if decode else S, kc, vc, 0, Tmax, 0, pos, vt)
        else:
            io = _lib.LlamaIO()
            io.x, io.h, io.qkv, io.att, io.act, io.u, io.u2 = (t.data_ptr() for t in (ws.x, ws.h, ws.qkv, ws.att, ws.act, ws.u, ws.u2))
            io.ldx, io.ldh, io.ldqkv, io.ldatt, io.ldact, io.ldu = (t.stride(0) for t in (ws.x, ws.h, ws.qkv, ws.att, ws.act, ws.u))
            io.route_ws, io.route_ws_bytes = ws.t.data_ptr(), ws.t.numel()
            sk = ops._splitk_workspace(dev)
            io.splitk_ws, io.splitk_ws_bytes = sk.data_ptr(), sk.numel()
            io.rope_tab = eng._rope_tab(Tmax).data_ptr()
            io.k_cache, io.v_cache, io.cache_layer_stride = kc.data_ptr(), vc.data_ptr(), kc.stride(0)
            if vt is not None:
                io.vt, io.vt_ld = vt.data_ptr(), vt.stride(-2)
            io.pos_dev = pos.data_ptr() if pos is not None else None
            io.B, io.S, io.Tmax, io.pos0 = B, 1 if decode else S, Tmax, 0
            io.x_fp32 = 1 if ws.x.dtype == torch.float32 else 0
            tab = eng._layer_table()
            fn = lib.crab_llama_layer_decode if decode else lib.crab_llama_layer_prefill
            for li in range(cfg.num_hidden_layers):
                _lib.check(fn(_lib.ctx(0), stream, C.byref(tab[li]), C.byref(io), li), 0)
            assert io.u_qkv_ready == 0                                   # the last layer has no next q|k|v group
        torch.cuda.synchronize()
        return ws.x[:M].clone(), ws.h[:M].clone()

    res = {}
    for per_layer in (False, True):
        kc = torch.zeros((cfg.num_hidden_layers, B, cfg.num_key_value_heads, 64, cfg.hidden_size // cfg.num_attention_heads), device="cuda", dtype=BF)
        vc = torch.zeros_like(kc)
        res[per_layer] = (run(per_layer, False, kc, vc), run(per_layer, True, kc, vc), kc.clone(), vc.clone())
    for i in range(2):
        for j in range(2):
            assert torch.equal(res[False][i][j], res[True][i][j]), (i, j)
    assert torch.equal(res[False][2], res[True][2]) and torch.equal(res[False][3], res[True][3])     # KV caches incl. the appended row
    assert res[False][0][0].float().abs().max() > 0
    # validation: a decode call with S != 1, a prefill call without vt, a group whose shapes do not chain
    tab = eng._layer_table()
    io = _lib.LlamaIO()
    assert lib.crab_llama_layer_decode(_lib.ctx(0), stream, C.byref(tab[0]), C.byref(io), 0) < 0
    assert b"llama_layer" in lib.crab_last_error(_lib.ctx(0))


def test_generate_splits_a_batch_that_does_not_fit_the_memory_budget():
    """Capacity planning: with a (faked) small memory budget generate() runs the batch as several groups one after the other and
    says so; ids are those of the one-piece run (rows are independent; per-step logits agree to the accumulation-order noise of
    the different M), first-step logits and per-step logits come back joined in row order."""
    meta, A = load_fixture("full_tiny_llama")
    model = build_tiny_crab(meta)
    model.load_state_dict(weights_from_table(meta), strict=False)
    um = model.base_model.model
    eng = um._engine
    g = torch.Generator(device="cuda").manual_seed(3)
    emb = A["embeds_bs2"][:1].cuda().to(BF).repeat(7, 1, 1)
    emb = emb + (torch.randn(emb.shape, device="cuda", generator=g) * 0.05).to(BF)
    kw = dict(eos_token_id=None, pad_token_id=2, return_step_logits=True, return_first_logits=True)
    ids, sl, fl = eng.generate(emb, 6, **kw)
    assert eng.last_plan["groups"] == [7]
    per = eng.bytes_per_sequence(emb.shape[1], 6)
    eng.kv_budget_bytes = int((eng.fixed_bytes(7, emb.shape[1]) + 3.5 * per) / 0.94) + 1
    try:
        with pytest.warns(RuntimeWarning, match="do not fit"):
            ids2, sl2, fl2 = eng.generate(emb, 6, **kw)
    finally:
        eng.kv_budget_bytes = None
    assert eng.last_plan["groups"] == [3, 2, 2]
    assert ids2.shape == ids.shape and sl2.shape == sl.shape and fl2.shape == fl.shape
    assert _rel(fl2, fl, "generate() split into groups vs one piece: first-step logits (HIP vs HIP)") < 6e-3
    assert _rel(sl2[:, 0], sl[:, 0], "generate() split into groups vs one piece: step-0 logits (HIP vs HIP)") < 6e-3
    same = (ids2 == ids).float().mean().item()
    assert same >= 0.9, same


def test_generate_with_sampling_like_the_reference_default():
    """generate(do_sample=True): HF sample mode with the defaults a Llama-2-chat checkpoint hands the reference's generate() call
    (temperature 0.6, top_k 50, top_p 0.9; scripts/quick_start.py:36-43): ids are valid, reproducible for a seed,
    different across seeds, identical between graph replay and eager, and top_k = 1 reproduces the greedy ids of the reference fixture."""
    meta, A = load_fixture("full_tiny_llama")
    model = build_tiny_crab(meta)
    model.load_state_dict(weights_from_table(meta), strict=False)
    mods = _inputs(meta)
    kw = dict(batch_input_ids=[A["ids0"]], batch_labels=[torch.full_like(A["ids0"], -100)], batch_X_modals=[mods[0]], batch_task_names=['avqa'],
              use_cache=True, max_new_tokens=meta["new_tokens"], pad_token_id=2, eos_token_id=None)
    a = model.generate(do_sample=True, seed=5, **kw)
    b = model.generate(do_sample=True, seed=5, **kw)
    c = model.generate(do_sample=True, seed=6, **kw)
    e = model.generate(do_sample=True, seed=5, use_graph=False, **kw)
    V = model.base_model.model.lm_head.weight.shape[0]
    assert a.shape == (1, meta["new_tokens"]) and int(a.min()) >= 0 and int(a.max()) < V
    assert torch.equal(a, b) and torch.equal(a, e) and not torch.equal(a, c)
    greedy = model.generate(do_sample=True, top_k=1, seed=5, **kw)
    assert torch.equal(greedy.cpu(), A["ids_bs1"])
    # without seed=: like HF, consecutive calls on the same prompt draw different streams; torch.manual_seed reproduces the pair
    torch.manual_seed(77)
    u1, u2 = model.generate(do_sample=True, **kw), model.generate(do_sample=True, **kw)
    torch.manual_seed(77)
    v1, v2 = model.generate(do_sample=True, **kw), model.generate(do_sample=True, **kw)
    assert not torch.equal(u1, u2) and torch.equal(u1, v1) and torch.equal(u2, v2)
    with pytest.raises(ValueError):
        model.generate(do_sample=True, temperature=0.0, **kw)


def test_generate_batches_in_flight_equals_separate_generate_calls():
    """generate_batches (several independent batches decoding on separate HIP streams, each with its own prepare_multimodal_inputs / left
    padding / KV cache / captured graph) returns for every batch exactly what a separate generate() call returns - greedy and sampled,
    ragged batch shapes (1 clip, 2 left-padded clips, 1 clip), EOS handling per batch."""
    meta, A = load_fixture("full_tiny_llama")
    model = build_tiny_crab(meta)
    model.load_state_dict(weights_from_table(meta), strict=False)
    mods = _inputs(meta)
    lab = [torch.full_like(A["ids0"], -100), torch.full_like(A["ids1"], -100)]
    batches = [dict(batch_input_ids=[A["ids0"]], batch_labels=[lab[0]], batch_X_modals=[mods[0]], batch_task_names=['avqa']),
               dict(batch_input_ids=[A["ids0"], A["ids1"]], batch_labels=lab, batch_X_modals=mods, batch_task_names=['avqa', 'avqa']),
               dict(batch_input_ids=[A["ids1"]], batch_labels=[lab[1]], batch_X_modals=[mods[1]], batch_task_names=['avqa'])]
    for kw in (dict(max_new_tokens=meta["new_tokens"], pad_token_id=2, eos_token_id=None),
               dict(max_new_tokens=meta["new_tokens"], pad_token_id=2, eos_token_id=None, do_sample=True, seed=3, top_k=8),
               dict(max_new_tokens=40, pad_token_id=2, eos_token_id=int(A["ids_bs1"][0, 3]))):
        many = model.generate_batches(batches, use_cache=True, **kw)
        assert len(many) == 3
        for g, b in enumerate(batches):
            kw1 = dict(kw)
            if kw1.get("do_sample"):
                kw1["seed"] = 3 + 7919 * g                        # batch g of generate_batches draws from its own stream
            solo = model.generate(**b, use_cache=True, **kw1)
            assert many[g].shape == solo.shape and torch.equal(many[g], solo), (g, kw)
    assert torch.equal(many[0][:, :4].cpu(), A["ids_bs1"][:, :4]) and many[0].shape[1] == 4     # stopped at the EOS it was given


def test_generate_many_refuses_batches_that_do_not_fit_together_and_returns_first_logits():
    """Engine level: generate_many with return_first_logits gives (ids, first-step logits) per batch, equal to generate()'s; with a (faked)
    small memory budget it raises MemoryError instead of failing in the allocator (the caller puts fewer batches in flight)."""
    meta, A = load_fixture("full_tiny_llama")
    model = build_tiny_crab(meta)
    model.load_state_dict(weights_from_table(meta), strict=False)
    eng = model.base_model.model._engine
    e1 = A["embeds_bs2"][:1].cuda().to(BF)
    e2 = A["embeds_bs2"].cuda().to(BF)
    kw = dict(eos_token_id=None, pad_token_id=2)
    outs = eng.generate_many([e1, e2, e1], 5, return_first_logits=True, **kw)
    assert len(outs) == 3
    for emb, (ids, fl) in zip([e1, e2, e1], outs):
        ids1, fl1 = eng.generate(emb, 5, return_first_logits=True, **kw)
        assert torch.equal(ids, ids1) and torch.equal(fl, fl1)
    per = eng.bytes_per_sequence(e2.shape[1], 5)
    eng.kv_budget_bytes = int(per * 1.5)
    try:
        with pytest.raises(MemoryError, match="fewer in flight"):
            eng.generate_many([e1, e2, e1], 5, **kw)
    finally:
        eng.kv_budget_bytes = None
    assert eng.generate_many([], 5) == []


def _coalesce_setup(fixture="full_tiny_llama"):
    meta, A = load_fixture(fixture)
    model = build_tiny_crab(meta)
    model.load_state_dict(weights_from_table(meta), strict=False)
    mods = _inputs(meta)
    lab = [torch.full_like(A["ids0"], -100), torch.full_like(A["ids1"], -100)]
    batches = [dict(batch_input_ids=[A["ids0"]], batch_labels=[lab[0]], batch_X_modals=[mods[0]], batch_task_names=['avqa']),
               dict(batch_input_ids=[A["ids0"], A["ids1"]], batch_labels=lab, batch_X_modals=mods, batch_task_names=['avqa', 'avqa']),
               dict(batch_input_ids=[A["ids1"]], batch_labels=[lab[1]], batch_X_modals=[mods[1]], batch_task_names=['avqa'])]
    return meta, A, model, batches


@pytest.mark.parametrize("fixture", ["full_tiny_llama", "full_tiny_qwen"])
def test_coalesced_batches_match_the_reference_fixture_per_batch(fixture):
    """(full_tiny_qwen: the same through the Qwen2 decoder - grouped-query decode attention with a first visible key per row, q / k / v bias in
    the fused RoPE epilogues.)
    generate_batches(coalesce=True): three eval-loop batches of DIFFERENT prompt lengths (1 clip; 2 left-padded clips; 1 shorter clip) decode
    as one ragged batch - right-aligned in one KV cache, per-row rotary offset and first visible key (crab_llama_io.row_off) - and every batch
    must come out as its own generate() call does in the REFERENCE: ids and per-step logits against the reference-recorded fixture
    (full_tiny_llama bs 1 and left-padded bs 2: the pads of the bs-2 batch are attended, positions run from 0 per batch), graph replay ==
    plain launches bit for bit, the Python per-launch sequencer == the native one, and the public API returns the same ids."""
    from crab_amd import decoder
    meta, A, model, batches = _coalesce_setup(fixture)
    um = model.base_model.model
    eng = um._engine
    n = meta["new_tokens"]
    inputs = um.prepare_multimodal_inputs_many(batches)
    embeds = [d["inputs_embeds"] for d in inputs]
    assert len({e.shape[1] for e in embeds}) >= 2, "the batches must differ in prompt length for this test to mean anything"
    assert _rel(embeds[0], A["embeds_bs1"], "coalesced encoders: inputs_embeds of batch 0 (bs 1)") < PB.bound(f"{fixture}: inputs_embeds")
    assert _rel(embeds[1], A["embeds_bs2"], "coalesced encoders: inputs_embeds of batch 1 (left-padded bs 2)") < PB.bound(f"{fixture}: inputs_embeds")
    assert torch.equal(inputs[1]["position_ids"].cpu().long(), A["pos_bs2"].long()) and torch.equal(inputs[1]["attention_mask"].cpu().long(), A["mask_bs2"].long())
    kw = dict(eos_token_id=None, pad_token_id=2, coalesce=True, return_step_logits=True)
    # both prefill forms of a ragged wave: per group into the right-aligned cache (cache pointers advanced), and MERGED - one front-padded batch
    # under forward()'s left-pad mask + position_ids through the native sequencer (crab_llama_io.pos_ids / kv_start), what a real question set gets
    saved = decoder.RAGGED_PAD_MAX
    try:
        for pad_max, form in ((0.0, "per_group"), (0.5, "merged")):
            decoder.RAGGED_PAD_MAX = pad_max
            res = eng.generate_many(embeds, n, **kw)
            assert eng.last_ragged_prefill == form
            res_eager = eng.generate_many(embeds, n, use_graph=False, **kw)
            decoder.NATIVE_LAYERS = False
            try:
                res_py = eng.generate_many(embeds, n, use_graph=False, **kw)
            finally:
                decoder.NATIVE_LAYERS = True
            for (i1, l1), (i2, l2), (i3, l3) in zip(res, res_eager, res_py):
                assert torch.equal(i1, i2) and torch.equal(l1, l2), "HIP-graph replay of the ragged step differs from plain launches"
                assert torch.equal(i1, i3) and torch.equal(l1, l3), f"the Python per-launch sequence differs from the native one ({form} prefill)"
            for g, key in ((0, "bs1"), (1, "bs2")):
                ids, logits = res[g]
                err = _check_ids(ids, A[f"ids_{key}"], A[f"logits_{key}"], logits)
                assert err < PB.bound(f"{fixture}: end to end") * A[f"logits_{key}"].abs().max().item(), (g, form, err)
    finally:
        decoder.RAGGED_PAD_MAX = saved
    res = eng.generate_many(embeds, n, **kw)                    # the default rule (these lengths: merged)
    # the public API: ids only, and with output_first_logits the first position's logits
    pub = model.generate_batches(batches, coalesce=True, use_cache=True, max_new_tokens=n, pad_token_id=2, eos_token_id=None)
    pub2 = model.generate_batches(batches, coalesce=True, use_cache=True, max_new_tokens=n, pad_token_id=2, eos_token_id=None, output_first_logits=True)
    for g in range(3):
        assert torch.equal(pub[g], res[g][0]) and torch.equal(pub2[g][0], res[g][0]) and torch.equal(pub2[g][1], res[g][1][:, 0])
    # and against separate generate() calls of the same model (HIP vs HIP; other kernels at the other M): ids wherever the margin allows
    for g, b in enumerate(batches):
        solo = model.generate(**b, use_cache=True, max_new_tokens=n, pad_token_id=2, eos_token_id=None, output_logits=True, return_dict_in_generate=True)
        sl = torch.stack(solo.logits, 1).float().cpu()
        _check_ids(res[g][0], solo.sequences.cpu(), sl, res[g][1])
        assert _rel(res[g][1], sl, f"coalesced batch {g} vs its own generate() call, per-step logits (HIP vs HIP)") < 2 * PB.bound(f"{fixture}: end to end")


def test_coalesced_batches_of_different_lengths_vs_oracle():
    """Five batches of two unsearched clips each, every batch with its own pair of prompt lengths (so its own left padding AND its own offset
    inside the right-aligned cache), coalesced into one ragged decode batch of 10 rows and - with max_rows = 4 - into three waves; each batch
    against the golden-pinned oracle run on THAT batch alone.  Greedy ids exact wherever the oracle's top-2 margin exceeds twice the logit
    error, per-step logits within the computed end-to-end bound of the stack."""
    from crab_amd import synth
    from oracle import crab_oracle as O
    from tests.test_oracle_golden import _full_cfg
    meta, A = load_fixture("full_tiny_llama")
    W = weights_from_table(meta)
    model = build_tiny_crab(meta)
    model.load_state_dict(W, strict=False)
    um = model.base_model.model
    Wo = _bf(O.strip_peft_prefix(W))
    ocfg = _full_cfg(meta)
    p = meta["prompts"]
    n = 10
    batches, refs = [], []
    for j, c0 in enumerate((51, 63, 75, 87, 99)):
        nts = (p["n0"] + 2 * j, p["n1"] + (5 * j) % 7)
        ids = [synth.synth_prompt_ids(nt, ocfg.base_vocab, model.SPECIAL_TOKEN_2_IDS, seed=meta["seed"], clip=c) for nt, c in zip(nts, (c0, c0 + 1))]
        mods = [{'<video>': synth.synth_video(p["t_v"], seed=meta["seed"], clip=c), '<audio>': synth.synth_audio(p["t_a"], p["l_a"], seed=meta["seed"], clip=c)}
                for c in (c0, c0 + 1)]
        batches.append(dict(batch_input_ids=ids, batch_labels=[torch.full_like(i, -100) for i in ids], batch_X_modals=mods, batch_task_names=['avqa'] * 2))
        refs.append(O.generate(ids, mods, Wo, ocfg, n))
    embeds = [d["inputs_embeds"] for d in um.prepare_multimodal_inputs_many(batches)]
    assert len({e.shape[1] for e in embeds}) >= 4
    from crab_amd import decoder
    worst = 0.0
    saved = decoder.RAGGED_PAD_MAX
    try:
        for max_rows, pad_max in ((None, 0.5), (None, 0.0), (4, 0.5)):          # one wave merged / per group, three waves
            decoder.RAGGED_PAD_MAX = pad_max
            res = um._engine.generate_many(embeds, n, eos_token_id=None, pad_token_id=2, coalesce=True, return_step_logits=True, max_rows=max_rows)
            assert um._engine.last_plan["groups"] == ([10] if max_rows is None else [4, 4, 2]) and um._engine.last_plan["coalesced"]
            assert um._engine.last_ragged_prefill == ("merged" if pad_max else "per_group")
            for (ids, logits), (ref_ids, ref_logits) in zip(res, refs):
                assert ids.shape == (2, n)
                worst = max(worst, _check_ids(ids, ref_ids, ref_logits, logits, min_frac=0.9) / ref_logits.abs().max().item())
    finally:
        decoder.RAGGED_PAD_MAX = saved
    assert worst < PB.bound("full_tiny_llama: end to end"), worst


def test_coalesced_batches_stop_per_batch_like_separate_calls():
    """EOS inside a coalesced wave: a batch whose rows have all finished returns what its own generate() call returns (trimmed at the column
    where its LAST row finished, finished rows padded) although the wave keeps stepping for the other batches."""
    meta, A, model, batches = _coalesce_setup()
    eos = int(A["ids_bs1"][0, 3])                                 # batch 0 (bs 1) hits it at its 4th token
    kw = dict(use_cache=True, max_new_tokens=24, pad_token_id=2, eos_token_id=eos)
    many = model.generate_batches(batches, coalesce=True, **kw)
    assert many[0].shape[1] == 4 and torch.equal(many[0][:, :4].cpu(), A["ids_bs1"][:, :4])
    for g, b in enumerate(batches):
        solo = model.generate(**b, **kw)
        assert many[g].shape == solo.shape, (g, many[g].shape, solo.shape)
        assert torch.equal(many[g], solo), g                      # fixture clips: wide margins, the ids agree across kernel regimes
    # min_new_tokens suppresses the EOS for every batch of the wave alike
    many2 = model.generate_batches(batches, coalesce=True, min_new_tokens=6, **kw)
    assert many2[0].shape[1] >= 6 and torch.equal(many2[0][:, :3], many[0][:, :3])


def test_harness_run_inference_coalesced(tmp_path):
    """harness.run_inference(coalesce=True): the eval loop's batches are collected up to `coalesce_rows` clips and decoded as ragged waves;
    the records (order, metadata, decoded text) equal the one-batch-at-a-time loop's."""
    from crab_amd import harness
    meta, A, model, batches = _coalesce_setup()

    class Tok:
        def batch_decode(self, ids, skip_special_tokens=False):
            return [" ".join(str(int(t)) for t in row) for row in ids]
    def with_meta(bs):
        out = []
        for i, b in enumerate(bs):
            d = dict(b)
            d["batch_metadata"] = [{"instruction": f"q{i}.{j}", "output": "x"} for j in range(len(b["batch_input_ids"]))]
            out.append(d)
        return out
    loop = batches + batches[:2]                                   # 5 batches, 7 clips
    kw = dict(max_new_tokens=meta["new_tokens"], pad_token_id=2, eos_token_id=None)
    a = harness.run_inference(with_meta(loop), model, Tok(), **kw)
    b = harness.run_inference(with_meta(loop), model, Tok(), coalesce=True, coalesce_rows=4, **kw)     # waves of >= 4 clips, then the rest
    c = harness.run_inference(with_meta(loop), model, Tok(), coalesce=True, in_flight=5, **kw)
    assert [r["instruction"] for r in a] == [r["instruction"] for r in b] == [r["instruction"] for r in c] and len(a) == 7
    assert [r["predict"] for r in a] == [r["predict"] for r in b] == [r["predict"] for r in c]


@pytest.mark.parametrize("fixture", ["full_tiny_llama", "full_tiny_qwen"])
def test_prefill_last_rows_only_equals_the_full_last_layer(fixture):
    """generate()'s prefill runs the LAST layer's attention / o_proj / MLP for the last row of every sequence only (crab_llama_io.last_rows_only:
    lm_head reads one row per sequence, the reference computes all S and drops S - 1, modeling_llama.py:1260).  Against the full last layer
    (CRAB_PREFILL_LAST_ROWS=0 = decoder.LAST_ROWS_ONLY False): the same ids, the first-step logits within what another kernel regime of the same
    rows costs (the B last rows go through the decode-regime GEMMs and the one-row attention instead of the prefill tiles), the KV cache of the
    last layer bit-identical (it is written by the same q|k|v projection), and both against the reference-recorded fixture."""
    from crab_amd import decoder
    meta, A, model, batches = _coalesce_setup(fixture)
    um = model.base_model.model
    eng = um._engine
    n = meta["new_tokens"]
    emb = um.prepare_multimodal_inputs(**batches[1])["inputs_embeds"]            # the left-padded batch of 2
    outs = {}
    for on in (True, False):
        decoder.LAST_ROWS_ONLY = on
        try:
            eng.invalidate()
            ids, logits = eng.generate(emb, n, eos_token_id=None, pad_token_id=2, return_step_logits=True)
            kc = eng._kv[(0,)][0][-1].clone()                                     # last layer's K cache
            outs[on] = (ids.cpu(), logits.float().cpu(), kc[:, :, : emb.shape[1]].clone())
        finally:
            decoder.LAST_ROWS_ONLY = True
    assert torch.equal(outs[True][0], outs[False][0])
    assert torch.equal(outs[True][2], outs[False][2]), "the last layer's prompt K rows differ"
    assert _rel(outs[True][1], outs[False][1], f"{fixture}: last-rows-only prefill vs the full last layer, per-step logits (HIP vs HIP)") < 2 * PB.bound(f"{fixture}: end to end")
    for on in (True, False):
        err = _check_ids(outs[on][0], A["ids_bs2"], A["logits_bs2"], outs[on][1])
        assert err < PB.bound(f"{fixture}: end to end") * A["logits_bs2"].abs().max().item(), (on, err)


def test_one_token_shortcut_takes_the_fused_attention_by_shape_not_by_engine_state():
    """r06 (scripts/fuzz_engine_state.py): forward()'s one-token shortcut runs in the engine's SHARED, grow-only prefill workspace.  Whether its attention
    took the fused RoPE + append + split-context launch used to depend on how large an earlier prefill had made that workspace (its scratch was sized - or
    omitted - for the workspace's rows, not the call's), and a 4-row step followed by a 2-row step in one scratch would have found partials where the 2-row
    launch keeps its zeroed arrival counters.  Now: the same call gives the same bits whatever ran before, and the launch trace shows the fused kernel."""
    import os, runpy
    from crab_amd import ops
    ns = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_engine_state.py"), run_name="lib")
    model = ns["build"](False)
    um = model.base_model.model
    eng = um._engine
    hid = um.config.hidden_size
    g = torch.Generator().manual_seed(5)
    mk = lambda B, S: (torch.randn(B, S, hid, generator=g) * 0.5).to(BF).cuda()
    e2, e4, big = mk(2, 40), mk(4, 5), mk(65, 33)

    def call(emb, steps=2):
        o = um(inputs_embeds=emb, use_cache=True)
        outs, pkv, lg = [o.logits.clone()], o.past_key_values, o.logits
        for _ in range(steps):
            st = um(input_ids=lg[:, -1].argmax(-1)[:, None], past_key_values=pkv)
            outs.append(st.logits.clone())
            pkv, lg = st.past_key_values, st.logits
        return outs

    eng.invalidate()
    fresh2 = call(e2)
    eng.invalidate()
    fresh4 = call(e4)
    eng.invalidate()
    eng.generate(big, 3, eos_token_id=None, pad_token_id=2)                  # the prefill workspace is now 2145 rows
    with ops.launch_trace() as tr:
        after_big = call(e2)
    assert tr.launched("attn_decode_rope_kernel<128>") >= 2 * len(um.model.layers), tr.counts      # fused at 2 rows x 2 heads, as in the fresh state
    assert all(torch.equal(a, b) for a, b in zip(after_big, fresh2))
    eng.invalidate()
    call(mk(4, 17))                                                          # a small shared workspace (68 rows) that both calls below run in
    a4, a2, b4 = call(e4), call(e2), call(e4)                                # 4-row steps, then 2-row steps, then 4 again: one scratch per row count
    assert all(torch.equal(a, b) for a, b in zip(a4, fresh4)) and all(torch.equal(a, b) for a, b in zip(b4, fresh4))
    assert all(torch.equal(a, b) for a, b in zip(a2, fresh2))


def test_generate_argument_edges_follow_hf_or_say_why_not():
    """max_new_tokens < 1 is HF's ValueError (GenerationConfig.validate), not a write past the id buffer; a one-element eos list is that id; a list of
    several stop ids - HF accepts one - is refused by name (the device-resident loop carries one id)."""
    import os, runpy
    ns = runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_engine_state.py"), run_name="lib")
    model = ns["build"](False)
    eng = model.base_model.model._engine
    emb = (torch.randn(3, 5, model.base_model.model.config.hidden_size) * 0.5).to(BF).cuda()
    for n in (0, -2):
        with pytest.raises(ValueError, match="max_new_tokens"):
            eng.generate(emb, n, eos_token_id=None, pad_token_id=2)
        with pytest.raises(ValueError, match="max_new_tokens"):
            eng.generate_many([emb, emb[:1]], n, eos_token_id=None, pad_token_id=2, coalesce=True)
    free = eng.generate(emb, 6, eos_token_id=None, pad_token_id=2)
    eos = int(free[1, 2])
    a = eng.generate(emb, 6, eos_token_id=eos, pad_token_id=2)
    b = eng.generate(emb, 6, eos_token_id=[eos], pad_token_id=2)
    assert torch.equal(a, b) and a.shape[1] <= 6 and int(a[1, 2]) == eos and (a.shape[1] == 3 or bool((a[1, 3:] == 2).all()))
    with pytest.raises(NotImplementedError, match="ONE id"):
        eng.generate(emb, 6, eos_token_id=[eos, eos + 1], pad_token_id=2)
