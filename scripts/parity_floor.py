"""The bf16-OPERAND FLOOR next to the storage emulation, on the CPU (no GPU needed): for every component of the path the oracle is run three
ways on the reference-recorded tiny fixtures -
    fp32      the reference arithmetic (what the fixtures hold),
    floor     emulate=O.OPERANDS: only matrix operands rounded to bf16, once (weights, linear-layer inputs, q / k / v) - the best ANY bf16-MFMA
              implementation can do,
    storage   emulate=torch.bfloat16 on bf16-rounded weights: every storage point of the HIP path rounded -
and max|x - fp32| / max|fp32| is printed for the last two.  floor = irreducible; storage - floor = what removable storage points cost.
    python scripts/parity_floor.py [--json out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import crab_oracle as O
from tests.util import load_fixture, weights_from_table
from tests.test_oracle_golden import _beats_cfg, _full_cfg, _full_inputs

BF = torch.bfloat16


def bfw(W):
    from tests.util import stored_params
    return stored_params(W)              # what the HIP modules hold: bf16, the encoders' LayerNorm parameters fp32


def q(x, e):
    """An INPUT of the component as the execution mode `e` receives it: the fp32 reference (e is None) gets the fixture's own fp32 values, every bf16
    execution - floor, storage emulation, and the HIP entry points, whose activations are bf16 - their bf16 rounding.  (r06: the reference side used
    to be rounded as well, which left the input quantisation out of the floors while the HIP path was compared with the unrounded fixture.)"""
    if isinstance(x, dict):
        return {k: q(v, e) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [q(v, e) for v in x]
    return x if e is None else x.to(BF).float()


def rel(a, b):
    return float((a.float() - b.float()).abs().max() / b.float().abs().max())


def rows(hip=None):
    """hip: optional {component name: callable -> tensor or list of tensors}: the HIP path's output for the same component (tests/
    test_parity_floor.py supplies it on the GPU box); its error against the same fp32 result becomes the row's "hip" field."""
    out = []

    def three(name, fn, W):
        ref = fn(W, None)
        flo = fn(W, O.OPERANDS)
        sto = fn(bfw(W), BF)
        got = hip[name]() if hip and name in hip else None
        if not isinstance(ref, (list, tuple)):
            ref, flo, sto = [ref], [flo], [sto]
            got = [got] if got is not None else None
        for i, (r, f, s_) in enumerate(zip(ref, flo, sto)):
            row = {"what": name if len(ref) == 1 else f"{name} [{i}]", "floor": rel(f, r), "storage_emulation": rel(s_, r), "scale": float(r.abs().max())}
            if got is not None:
                row["hip"] = rel(got[i].detach().float().cpu().reshape(r.shape), r)
            out.append(row)

    meta, A = load_fixture("clip_tiny")
    from crab_amd import synth
    cfg = O.ClipConfig(**meta["cfg"], select_layers=tuple(meta["select"]))
    video = synth.synth_video(meta["t_v"], seed=meta["seed"], clip=meta["clip"])[None]
    three("clip_tiny feature levels", lambda W, e: O.visual_encoder(q(video, e), W, cfg, e), weights_from_table(meta))

    meta, A = load_fixture("beats_tiny")
    bc = _beats_cfg(meta["cfg"])
    for L in (98, 198):
        three(f"beats_tiny L={L}", lambda W, e, L=L: O.beats(q(A[f"x{L}"], e), W, bc, emulate=e), weights_from_table(meta))

    meta, A = load_fixture("projectors_tiny")
    qf = O.QFormerConfig(hidden_size=meta["qf"]["hidden"], num_attention_heads=meta["qf"]["heads"], intermediate_size=meta["qf"]["inter"])
    W = weights_from_table(meta)
    three("VLProjector (tiny)", lambda W, e: O.vl_projector(q(A["vfeat"], e), W, qf, emulate=e), W)
    three("ALProjector (tiny)", lambda W, e: O.al_projector(q(A["afeat"], e), W, qf, emulate=e), W)

    for fx in ("full_tiny_llama", "full_tiny_qwen"):
        meta, A = load_fixture(fx)
        W = O.strip_peft_prefix(weights_from_table(meta))
        cfg = _full_cfg(meta)
        mods = _full_inputs(meta)
        three(f"{fx}: inputs_embeds bs2 (encoders + projectors + splice)",
              lambda W, e: O.prepare_multimodal_inputs([A["ids0"], A["ids1"]], q(mods, e), W, cfg, e)["inputs_embeds"], W)
        three(f"{fx}: decoder prefill logits, all rows (from the reference's inputs_embeds)",
              lambda W, e: O.decoder_forward(q(A["embeds_bs1"], e), W, cfg.decoder, emulate=e)[0], W)
        n = meta["new_tokens"]

        def gen(W, e):          # teacher-forced on the reference's ids: per-step logits of the SAME contexts
            emb = O.prepare_multimodal_inputs([A["ids0"]], q(mods[:1], e), W, cfg, e)["inputs_embeds"]
            toks = W["model.embed_tokens.weight"].float()[A["ids_bs1"][0, :n - 1]][None]
            logits, _, _ = O.decoder_forward(torch.cat([emb, O._r(toks, e)], 1), W, cfg.decoder, emulate=e)
            return logits[:, -n:]
        three(f"{fx}: end to end, per-step logits of {n} teacher-forced greedy steps", gen, W)
    # ---- forward() under masks (unified_llama.py:129-160 fixtures): logits of the defined rows + the one-token shortcut on the kept cache
    meta, A = load_fixture("forward_masked_tiny_llama")
    W = O.strip_peft_prefix(weights_from_table(meta))
    dcfg = _full_cfg(meta).decoder
    valid = A["mask_bs2"].bool()

    def masked(W, e):
        logits, hn, cache = O.decoder_forward(q(A["embeds_bs2"], e), W, dcfg, positions=A["pos_bs2"], attention_mask=A["mask_bs2"], emulate=e)
        tok = W["model.embed_tokens.weight"].float()[A["step_tok"]][:, None]
        l2, _, _ = O.decoder_forward(O._r(tok, e), W, dcfg, cache, positions=A["step_pos"], attention_mask=A["step_mask"], emulate=e)
        return [logits[valid], hn[valid], l2]
    three("forward_masked_tiny_llama: left-pad mask + position_ids (valid logits | post-norm hidden | decode-shortcut logits)", masked, W)
    mcfg = _full_cfg(meta)
    mmods = _full_inputs(meta)

    def multimodal(W, e):          # the multimodal branch of forward(): encoders -> splice -> left pad -> decoder under the mask; logits of EVERY valid row
        inp = O.prepare_multimodal_inputs([A["ids0"], A["ids1"]], q(mmods, e), W, mcfg, e)
        logits, _, _ = O.decoder_forward(inp["inputs_embeds"], W, dcfg, positions=inp["position_ids"], attention_mask=inp["attention_mask"], emulate=e)
        return logits[valid]
    three("multimodal forward_masked_tiny_llama: forward(batch_input_ids=...) logits of all valid rows", multimodal, W)
    meta, A = load_fixture("forward_holes_tiny_llama")
    W = O.strip_peft_prefix(weights_from_table(meta))
    hcfg = O.DecoderConfig(**meta["dec"])
    seen = A["mask"].cumsum(-1) > 0

    def holes(W, e):
        logits, hn, cache = O.decoder_forward(q(A["embeds"], e), W, hcfg, attention_mask=A["mask"], emulate=e)
        lp, _, _ = O.decoder_forward(q(A["embeds"], e), W, hcfg, positions=A["pos"], attention_mask=A["mask"], emulate=e)
        tok = W["model.embed_tokens.weight"].float()[A["step_tok"]][:, None]
        l2, _, _ = O.decoder_forward(O._r(tok, e), W, hcfg, cache, positions=A["step_pos"], attention_mask=A["step_mask"], emulate=e)
        return [logits[seen], hn[seen], lp[seen], l2]
    three("forward_holes_tiny_llama: mask with interior holes (logits | hidden | logits under cumsum positions | decode-shortcut logits)", holes, W)
    # ---- one layer of the reference's vendored modeling files: prefill + cached decode step (layer output = the residual stream)
    for fx in ("llama_ops", "qwen_ops"):
        meta, A = load_fixture(fx)
        W = weights_from_table(meta)
        c = dict(meta["cfg"])
        lcfg = O.DecoderConfig(**{**c, "num_hidden_layers": 1, "vocab_size": 320})
        S = A["layer_x"].shape[1]

        def layer(W, e, A=A, lcfg=lcfg, S=S):
            cache = O.KVCache()
            y = O.decoder_layer(q(A["layer_x"], e), W, 0, lcfg, cache, torch.arange(S)[None], emulate=e)
            y1 = O.decoder_layer(q(A["layer_x1"], e), W, 0, lcfg, cache, torch.tensor([[S]]), emulate=e)
            return [y, y1]
        three(f"{fx}: one hyper-LoRA decoder layer (prefill output | cached decode-step output)", layer, W)
    return out


if __name__ == "__main__":
    torch.manual_seed(0)
    R = rows()
    print(f"{'component':92s} {'floor':>9s} {'storage':>9s} {'ratio':>6s}")
    for r in R:
        print(f"{r['what'][:92]:92s} {r['floor']:9.2e} {r['storage_emulation']:9.2e} {r['storage_emulation'] / max(r['floor'], 1e-12):6.2f}"
              + (f"  hip {r['hip']:9.2e}" if "hip" in r else ""))
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(R, f, indent=1)
