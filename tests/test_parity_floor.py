"""How close can ANY bf16-MFMA implementation of this path come to the fp32 reference, and where does the HIP path stand against that?

north_star asks for "logits within 1e-3 bf16".  The oracle's operand-floor mode (oracle/crab_oracle.py: emulate=O.OPERANDS) rounds ONLY what a
matrix instruction must consume in bf16 - weights, linear-layer inputs, q / k / v - once, and keeps everything else in fp32: no storage format
does better.  The CPU test pins the statement "1e-3 is below that floor" on the reference-recorded fixtures; the GPU test records, per component,
{floor, storage emulation, HIP} against the same fp32 result (profiles/r05_parity_report.json) and requires the HIP path to stay within 1.5x of
the larger of the two emulations - i.e. at the level of its storage format, which since r05 (fp32 LayerNorm parameters in the encoders) sits
at 1.1-1.9x the floor."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
BF = torch.bfloat16


def _rows(hip=None):
    import parity_floor
    return parity_floor.rows(hip)


def test_bf16_operand_floor_is_above_the_1e3_the_north_star_names():
    """CPU only.  On every decoder row of the tiny fixtures (2-layer hyper-LoRA Llama / Qwen2 stacks: prefill logits of all rows, end-to-end
    per-step logits) rounding nothing but the MFMA operands to bf16 already moves the logits by 2.9e-3 ... 4.0e-3 of their scale; the
    encoder features by 3.5e-3 ... 6.9e-3.  A tolerance of 1e-3 against the fp32 reference is therefore unreachable with bf16 matrix
    operands whatever the kernels do; the storage emulation (what the HIP path's storage points cost on top) stays within 2.1x of the floor."""
    R = [r for r in _rows() if not any(k in r["what"] for k in ("_ops:", "forward_masked", "forward_holes"))]      # the component rows of r05 (the r06
    assert len(R) >= 13                                             # rows of the masked forwards / single layers are bounds for tests/test_model_gpu.py)
    for r in R:
        assert r["floor"] > 1.0e-3, r                               # measured: >= 2.88e-3 on every row
        assert r["storage_emulation"] < 2.5 * r["floor"], r        # measured: <= 2.05 (beats_tiny L=198)
    dec = [r for r in R if "logits" in r["what"]]
    assert len(dec) == 4 and min(r["floor"] for r in dec) > 2.5e-3
    extra = [r for r in _rows() if r not in R]
    assert len(extra) >= 12 and all(r["floor"] > 5e-4 and r["storage_emulation"] < 2.5 * r["floor"] for r in extra), extra


@pytest.mark.gpu
def test_hip_path_against_the_operand_floor_and_the_storage_emulation():
    from crab_amd import ops, synth
    from crab_amd.multimodal_encoder import ALProjector, AudioEncoder, VisualEncoder, VLProjector
    from tests.util import bert_cfg, build_tiny_crab, load_fixture, record_parity, weights_from_table
    hip = {}
    meta, A = load_fixture("clip_tiny")
    W = weights_from_table(meta)
    ve = VisualEncoder(select_layer_list=meta["select"], config=meta["cfg"], device="cuda")
    ve.load_state_dict({k[len("model.visual_encoder."):]: v for k, v in W.items()}, strict=False)
    video = synth.synth_video(meta["t_v"], seed=meta["seed"], clip=meta["clip"])[None]
    hip["clip_tiny feature levels"] = lambda: ve(ops.cast_bf16(video.cuda()))
    metab, Ab = load_fixture("beats_tiny")
    ae = AudioEncoder(cfg=metab["cfg"], device="cuda")
    ae.load_state_dict({k[len("model.audio_encoder."):]: v for k, v in weights_from_table(metab).items()}, strict=False)
    for L in (98, 198):
        hip[f"beats_tiny L={L}"] = (lambda L=L: ae(ops.cast_bf16(Ab[f"x{L}"].cuda())))
    metap, Ap = load_fixture("projectors_tiny")
    Wp = weights_from_table(metap)
    bc = bert_cfg(metap["qf"])
    vl = VLProjector(hidden_size=128, image_token_nums=256, num_query_token=32, num_hidden_layers=2, d_model=metap["d_model"], depth=2, bert_config=bc, device="cuda")
    vl.load_state_dict({k[len("model.vl_projector."):]: v for k, v in Wp.items() if k.startswith("model.vl_projector.")}, strict=False)
    al = ALProjector(hidden_size=128, num_query_token=32, num_hidden_layers=2, d_model=metap["d_model"], depth=2, bert_config=bc, device="cuda")
    al.load_state_dict({k[len("model.al_projector."):]: v for k, v in Wp.items() if k.startswith("model.al_projector.")}, strict=False)
    hip["VLProjector (tiny)"] = lambda: vl(Ap["vfeat"].to(BF).cuda())
    hip["ALProjector (tiny)"] = lambda: al(Ap["afeat"].to(BF).cuda())
    keep = []
    for fx in ("full_tiny_llama", "full_tiny_qwen"):
        m, F_ = load_fixture(fx)
        model = build_tiny_crab(m)
        model.load_state_dict(weights_from_table(m), strict=False)
        keep.append(model)
        p = m["prompts"]
        mods = [{'<video>': synth.synth_video(p["t_v"], seed=m["seed"], clip=c), '<audio>': synth.synth_audio(p["t_a"], p["l_a"], seed=m["seed"], clip=c)}
                for c in (p["clip0"], p["clip1"])]
        lab = [torch.full_like(F_["ids0"], -100), torch.full_like(F_["ids1"], -100)]
        n = m["new_tokens"]
        hip[f"{fx}: inputs_embeds bs2 (encoders + projectors + splice)"] = (
            lambda model=model, F_=F_, mods=mods, lab=lab: model.prepare_multimodal_inputs([F_["ids0"], F_["ids1"]], lab, mods, ['avqa', 'avqa'])["inputs_embeds"])
        hip[f"{fx}: decoder prefill logits, all rows (from the reference's inputs_embeds)"] = (
            lambda model=model, F_=F_: model.base_model.model(inputs_embeds=F_["embeds_bs1"].to(BF).cuda()).logits)

        def gen(model=model, F_=F_, mods=mods, lab=lab, n=n):
            r = model.generate(batch_input_ids=[F_["ids0"]], batch_labels=lab[:1], batch_X_modals=mods[:1], batch_task_names=['avqa'], use_cache=True,
                               max_new_tokens=n, pad_token_id=2, eos_token_id=None, output_logits=True, return_dict_in_generate=True)
            assert torch.equal(r.sequences.cpu(), F_["ids_bs1"]), "the fixture clips decode to the reference's ids (same contexts at every step)"
            return torch.stack(r.logits, 1)
        hip[f"{fx}: end to end, per-step logits of {n} teacher-forced greedy steps"] = gen
    R = _rows(hip)
    worst = 0.0
    for r in R:
        if "hip" not in r:                                          # the masked-forward / single-layer rows: bounded in tests/test_model_gpu.py through tests/bounds.py
            assert any(k in r["what"] for k in ("_ops:", "forward_masked", "forward_holes")), r["what"]
            continue
        record_parity("TRIPLET " + r["what"], r["hip"] * r["scale"], r["scale"], None, floor=r["floor"], storage_emulation=r["storage_emulation"],
                      hip=r["hip"], hip_over_floor=r["hip"] / r["floor"], hip_over_storage_emulation=r["hip"] / r["storage_emulation"])
        # the HIP path sits at the level of its storage format (accumulation order and 1-ulp flips on top): never beyond 1.5x the larger emulation
        assert r["hip"] <= 1.5 * max(r["storage_emulation"], r["floor"]), r
        worst = max(worst, r["hip"] / r["floor"])
    assert worst < 3.0, worst
