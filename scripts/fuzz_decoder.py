"""Differential fuzz of the whole decoder path (prefill chunks, KV cache, the decode regimes M <= 16 / <= 128 / <= 256 / <= 512, HIP-graph replay and
eager, the C layer sequencer and the per-launch Python one, hyper-LoRA of every projection, GQA, q|k|v bias) on RANDOM tiny configurations against the
fp32 CPU oracle: greedy ids where the reference margin allows, per-step logits within the decoder tolerance, the two sequencers bit-identical.
    python scripts/fuzz_decoder.py [configs] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd import decoder, ops
from crab_amd.peft_hyper import LoraConfig, get_peft_model
from oracle import crab_oracle as O

BF = torch.bfloat16
NCFG = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
REL = 1.2e-2
bad = []
worst_all = 0.0
WIDE = os.environ.get("CRAB_FUZZ_WIDE") == "1"      # full-width rows (Llama-2-7B / Qwen2-7B projection shapes, 2 layers): the kernels of the benchmark regimes
for ci in range(NCFG):
    qwen = rng.random() < 0.4
    d = rng.choice([64, 128])
    Hk = rng.choice([1, 2, 4])
    G = rng.choice([1, 1, 2, 4]) if not qwen else rng.choice([1, 2, 7])
    H = Hk * G
    if H * d > 1024: continue
    hid = H * d
    inter = rng.choice([64, 136, 352, 1000, 2 * hid + 8])
    L = rng.choice([1, 2, 3])
    V = rng.choice([320, 515, 1000])
    r, nl = rng.choice([(8, 3), (8, 3), (4, 2), (16, 3), (4, 8)])
    if WIDE:
        d, L, (r, nl) = 128, 2, (8, 3)
        if qwen: H, Hk, hid, inter, V = 28, 4, 3584, 18944, 4000
        else: H, Hk, hid, inter, V = 32, 32, 4096, 11008, 32017
    if qwen:
        from crab_amd.unified_qwen import UnifiedConfig, UnifiedForCausalLM
    else:
        from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    kw = dict(hidden_size=hid, intermediate_size=inter, num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hk, vocab_size=V,
              rms_norm_eps=rng.choice([1e-5, 1e-6]), rope_theta=rng.choice([1e4, 1e6]))
    torch.manual_seed(100 + ci)
    cfg = UnifiedConfig(**kw, pad_token_id=2, **({"attention_bias": True} if qwen else {}))
    um = UnifiedForCausalLM(cfg, device="cuda")
    model = get_peft_model(um, LoraConfig(r=r, lora_alpha=2 * r, lora_nums=nl))
    for n_, p in model.named_parameters():
        small = 0.2 if ("o_proj" in n_ or "down_proj" in n_ or "lora_B" in n_) else 1.0
        p.data.copy_((torch.randn(p.shape) * (1.4 / hid ** 0.5) * small).to(BF) if p.dim() > 1 else
                     ((1 + 0.1 * torch.randn(p.shape)) if "norm" in n_ else 0.1 * torch.randn(p.shape)).to(BF))
    W = {k: v.detach().float().cpu() for k, v in O.strip_peft_prefix(model.state_dict()).items() if v.dtype.is_floating_point}
    ocfg = O.DecoderConfig(**kw, lora_r=r, lora_alpha=2 * r, lora_nums=nl)
    eng = model.base_model.model._engine
    regimes = [(rng.choice([1, 2, 5]), rng.choice([1, 6, 33])), (rng.choice([16, 17, 40]), rng.choice([3, 9])), (rng.choice([130, 260]), 4)]
    if WIDE: regimes = [(rng.choice([1, 8, 16]), 5), (rng.choice([17, 100, 129]), 3), (rng.choice([255, 257, 300, 383]), 2), (rng.choice([448, 511, 512, 513, 600]), 2)]
    for B, S in regimes:
        n = 3
        desc = f"cfg {ci}: {'qwen' if qwen else 'llama'} hid={hid} H={H}/{Hk} d={d} I={inter} L={L} V={V} r={r} nl={nl} B={B} S={S}"
        emb = (torch.randn(B, S, hid) * 0.5).to(BF).cuda()
        outs = []
        try:
            for native in (True, False):
                decoder.NATIVE_LAYERS = native
                for use_graph in ((True, False) if native else (True,)):
                    eng._dec.clear()
                    rr = eng.generate(emb, n, eos_token_id=None, pad_token_id=2, return_step_logits=True, use_graph=use_graph)
                    outs.append((rr[0].clone().cpu(), rr[1].float().cpu()))
        except Exception as e:      # noqa: BLE001
            bad.append(desc + f" -> {type(e).__name__}: {str(e)[:200]}"); continue
        finally:
            decoder.NATIVE_LAYERS = True
        for ids, lg in outs[1:]:
            if not (torch.equal(ids, outs[0][0]) and torch.equal(lg, outs[0][1])):
                bad.append(desc + " -> sequencers / graph-vs-eager differ"); break
        rows = list(range(B)) if B <= 8 else sorted(rng.sample(range(B), 6) + [0, B - 1])
        ref_ids, ref_lg = O.greedy_generate(emb[rows].float().cpu(), W, ocfg, n)
        # the same oracle with every stored activation rounded to bf16 (fp32 residual stream): what the storage format alone costs on THIS
        # configuration - tiny random models amplify it; the HIP path may not exceed max(REL, 2.5 x that)
        emu_ids, emu_lg = O.greedy_generate(emb[rows].float().cpu(), W, ocfg, n, emulate=BF)
        ids, lg = outs[0][0][rows], outs[0][1][rows]
        scale = float(ref_lg.abs().max())
        top2 = ref_lg.topk(2, -1).values
        margin = top2[..., 0] - top2[..., 1]
        emu_seen = 0.0
        for b in range(len(rows)):
            for s in range(n):
                err = float((lg[b, s] - ref_lg[b, s]).abs().max())
                # once the EMULATED run has left the reference's token sequence (a sub-noise margin) its logits say nothing about this context: the bound
                # falls back to the largest emulation error seen on this configuration's earlier rows / steps (at least 2 x REL)
                emu_ok = bool((emu_ids[b, :s] == ref_ids[b, :s]).all())
                emu = float((emu_lg[b, s] - ref_lg[b, s]).abs().max()) if emu_ok else max(emu_seen, 0.8 * REL * scale)
                if emu_ok: emu_seen = max(emu_seen, emu)
                worst_all = max(worst_all, err / scale)
                if err > max(REL * scale, 2.5 * emu):
                    bad.append(desc + f" -> row {rows[b]} step {s}: logit err {err / scale:.3e} of scale (bf16-storage emulation {emu / scale:.3e})"); break
                if ids[b, s] != ref_ids[b, s]:
                    if margin[b, s] > 2 * err: bad.append(desc + f" -> row {rows[b]} step {s}: id differs at margin {float(margin[b, s]):.4f} > 2 x err {err:.4f}")
                    break
        # ---- the EOS / min_new_tokens / pad state machine of the device-resident loop, checked against the run's OWN per-step logits (HF semantics,
        # EOS suppressed while step < min_new_tokens, a finished row emits pad, the loop ends once every row has finished)
        n2 = 6
        free = eng.generate(emb, n2, eos_token_id=None, pad_token_id=2, return_step_logits=True)[0].cpu()
        eos = int(free[rng.randrange(B), rng.randrange(1, n2)])                  # a token some row really produces
        mn = rng.choice([0, 0, 1, 3])
        try:
            eng._dec.clear()
            ids2, lg2 = eng.generate(emb, n2, eos_token_id=eos, pad_token_id=2, min_new_tokens=mn, return_step_logits=True)
        except Exception as e:      # noqa: BLE001
            bad.append(desc + f" eos={eos} min_new={mn} -> {type(e).__name__}: {str(e)[:200]}"); continue
        ids2, lg2 = ids2.cpu(), lg2.float().cpu()
        fin = torch.zeros(B, dtype=torch.bool)
        exp, steps = [], 0
        for s in range(ids2.shape[1]):
            l = lg2[:, s].clone()
            if s < mn: l[:, eos] = float("-inf")
            tok = l.argmax(-1)
            tok = torch.where(fin, torch.full_like(tok, 2), tok)
            exp.append(tok)
            fin = fin | (tok == eos)
            steps = s + 1
            if bool(fin.all()): break
        exp = torch.stack(exp, 1)
        want_len = steps if bool(fin.all()) else n2
        if ids2.shape[1] != want_len or not torch.equal(ids2[:, :exp.shape[1]], exp):
            bad.append(desc + f" eos={eos} min_new={mn} -> ids {tuple(ids2.shape)} do not follow the EOS / pad rules from their own logits (expected length {want_len})")
    # ---- forward() under random 2-D attention masks (left padding, interior holes, both) and optional cumsum-1 position_ids, then the one-token
    # decode shortcut on the kept cache: logits of every query row that sees at least one key against the oracle
    for _ in range(2):
        B, S = rng.choice([1, 2, 4]), rng.choice([5, 17, 40, 70])
        emb = (torch.randn(B, S, hid) * 0.5).to(BF).cuda()
        mask = (torch.rand(B, S) < rng.choice([0.6, 0.85, 1.0])).long()
        for b in range(B):
            mask[b, :rng.randrange(0, S // 2 + 1)] = 0 if rng.random() < 0.5 else mask[b, :1].item()
            mask[b, rng.randrange(S // 2, S)] = 1                                  # at least one visible key in the second half
        use_pos = rng.random() < 0.5
        pos = (mask.cumsum(-1) - 1).clamp(min=0) if use_pos else None
        desc = f"cfg {ci}: forward() {'qwen' if qwen else 'llama'} hid={hid} H={H}/{Hk} d={d} L={L} B={B} S={S} masked={int((mask == 0).sum())} pos_ids={use_pos}"
        try:
            out = model.base_model.model(inputs_embeds=emb, attention_mask=mask.cuda(), position_ids=pos.cuda() if use_pos else None, use_cache=True)
            tok = out.logits[:, -1].argmax(-1)
            mask2 = torch.cat([mask, torch.ones(B, 1, dtype=torch.long)], 1)
            pos2 = (pos[:, -1:] + 1) if use_pos else torch.full((B, 1), S, dtype=torch.long)
            step = model.base_model.model(input_ids=tok[:, None], attention_mask=mask2.cuda(), position_ids=pos2.cuda(), past_key_values=out.past_key_values)
        except Exception as e:      # noqa: BLE001
            bad.append(desc + f" -> {type(e).__name__}: {str(e)[:200]}"); continue
        seen = mask.cumsum(-1) > 0
        ref, _, cache = O.decoder_forward(emb.float().cpu(), W, ocfg, positions=pos, attention_mask=mask)
        emu, _, cache_e = O.decoder_forward(emb.float().cpu(), W, ocfg, positions=pos, attention_mask=mask, emulate=BF)
        scale = float(ref[seen].abs().max())
        e1, e1m = float((out.logits.float().cpu() - ref)[seen].abs().max()), float((emu - ref)[seen].abs().max())
        worst_all = max(worst_all, e1 / scale)
        if e1 > max(REL * scale, 2.5 * e1m): bad.append(desc + f" -> prefill logits err {e1 / scale:.3e} of scale (emulation {e1m / scale:.3e})")
        e_tok = W["model.embed_tokens.weight"][tok.cpu()][:, None]
        ref2, _, _ = O.decoder_forward(e_tok, W, ocfg, cache, positions=pos2, attention_mask=mask2)
        emu2, _, _ = O.decoder_forward(e_tok, W, ocfg, cache_e, positions=pos2, attention_mask=mask2, emulate=BF)
        e2, e2m = float((step.logits.float().cpu() - ref2).abs().max()), float((emu2 - ref2).abs().max())
        if e2 > max(REL * scale, 2.5 * e2m): bad.append(desc + f" -> decode-shortcut logits err {e2 / scale:.3e} of scale (emulation {e2m / scale:.3e}, prefill err {e1 / scale:.3e})")
    del model, um
    torch.cuda.empty_cache()
    print(f"cfg {ci} done ({'qwen' if qwen else 'llama'} hid={hid} H={H}/{Hk} d={d} I={inter} L={L} V={V} r={r} nl={nl}); failures so far {len(bad)}", flush=True)
print(f"worst logit error {worst_all:.3e} of scale; {len(bad)} failures")
for b_ in bad[:40]: print("FAIL", b_)
sys.exit(1 if bad else 0)
