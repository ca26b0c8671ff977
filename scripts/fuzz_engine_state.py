"""Stateful fuzz of the generation engine (r06): the other generators give every call a fresh engine state (`eng._dec.clear()`), so nothing exercised what a
long evaluation does - hundreds of calls of changing shape on ONE engine that keeps KV slots, decode workspaces, captured HIP graphs and a grow-only prefill
workspace between them (and evicts them lazily).  Here a random SEQUENCE of calls - generate() at batch sizes across the decode regimes with and without
EOS / min_new_tokens / hidden states / step logits / sample mode / two decode streams / a KV budget that forces groups, generate_batches in flight and coalesced (ragged
waves), forward(use_cache) + the one-token shortcut - runs twice on the same tiny model: once with the engine invalidated before every call (fresh state),
once straight through (carried state, shapes revisited with new data so that cached graphs and buffers are reused).  Every result must be BIT-IDENTICAL:
the kernels chosen depend on the shapes only, never on what the engine holds.   python scripts/fuzz_engine_state.py [calls] [seed]"""
import os, random, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crab_amd.peft_hyper import LoraConfig, get_peft_model

BF = torch.bfloat16
warnings.filterwarnings("ignore", category=RuntimeWarning)      # the planner's "running N groups" notes under the tight budgets below
_argv = sys.argv[1:] if __name__ == "__main__" else []          # (tests import build() from this file: run_name != "__main__")
NCALL = int(_argv[0]) if len(_argv) > 0 else 60
SEED = int(_argv[1]) if len(_argv) > 1 else 0
rng = random.Random(SEED)
bad = []


def build(qwen):
    if qwen:
        from crab_amd.unified_qwen import UnifiedConfig, UnifiedForCausalLM
        kw = dict(hidden_size=256, intermediate_size=352, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, vocab_size=515, attention_bias=True)
    else:
        from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
        kw = dict(hidden_size=256, intermediate_size=136, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=2, vocab_size=320)
    torch.manual_seed(7)
    um = UnifiedForCausalLM(UnifiedConfig(**kw, pad_token_id=2, rms_norm_eps=1e-5, rope_theta=1e4), device="cuda")
    model = get_peft_model(um, LoraConfig(r=8, lora_alpha=16, lora_nums=3))
    for n_, p in model.named_parameters():
        small = 0.2 if ("o_proj" in n_ or "down_proj" in n_ or "lora_B" in n_) else 1.0
        p.data.copy_((torch.randn(p.shape) * (1.4 / 16) * small).to(BF) if p.dim() > 1 else
                     ((1 + 0.1 * torch.randn(p.shape)) if "norm" in n_ else 0.1 * torch.randn(p.shape)).to(BF))
    return model


def make_calls(n, hid, V):
    """descriptions only (shapes, flags, data seeds): both passes build the same tensors from them"""
    calls = []
    for i in range(n):
        kind = rng.choice(["generate"] * 5 + ["batches", "batches", "forward"])
        c = {"kind": kind, "seed": 1000 + i}
        if kind == "generate":
            c.update(B=rng.choice([1, 2, 8, 16, 17, 40, 65, 130, 260]), S=rng.choice([1, 6, 33]), n=rng.choice([1, 2, 3, 5]),
                     graph=rng.random() < 0.8, streams=rng.choice([1, 1, 1, 2]), eos=rng.choice([None, None, "pick"]), min_new=rng.choice([0, 0, 2]),
                     extra=rng.choice([None, None, "hidden", "logits", "first"]), budget=rng.choice([None, None, None, "tight"]),
                     sampling=rng.choice([None, None, None, (0.7, 20, 0.9), (1.3, 0, 0.5), (0.6, 5, 1.0)]))
        elif kind == "batches":
            c.update(sizes=[rng.choice([1, 2, 3, 8]) for _ in range(rng.choice([1, 2, 3, 5]))], S=[rng.choice([2, 5, 9, 20]) for _ in range(5)],
                     n=rng.choice([1, 3, 4]), coalesce=rng.random() < 0.6, max_rows=rng.choice([None, None, 4, 9]))
        else:
            c.update(B=rng.choice([1, 2, 4]), S=rng.choice([5, 17, 40]))
        calls.append(c)
    return calls


def run(model, c, hid, V):
    um = model.base_model.model
    eng = um._engine
    g = torch.Generator().manual_seed(c["seed"])
    rn = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(BF).cuda()
    out = []
    if c["kind"] == "generate":
        emb = rn(c["B"], c["S"], hid)
        eos = None
        if c["eos"] == "pick":
            eos = int(torch.randint(3, V, (1,), generator=g))
        eng.kv_budget_bytes = None
        if c["budget"] == "tight" and c["B"] >= 8:
            eng.kv_budget_bytes = int(eng.fixed_bytes(c["B"], c["S"]) / 0.94 + 0.6 * c["B"] * eng.bytes_per_sequence(c["S"], c["n"]) / 0.94)      # ~2 groups
        kw = dict(eos_token_id=eos, pad_token_id=2, min_new_tokens=c["min_new"], use_graph=c["graph"], decode_streams=c["streams"])
        if c["sampling"] is not None: kw["sampling"] = c["sampling"] + (c["seed"],)          # HF sample mode (temperature, top_k, top_p, seed): the seed is baked into the graph
        if c["extra"] == "hidden": kw["return_hidden"] = True
        if c["extra"] == "logits": kw["return_step_logits"] = True
        if c["extra"] == "first": kw["return_first_logits"] = True
        try:
            r = eng.generate(emb, c["n"], **kw)
        finally:
            eng.kv_budget_bytes = None
        out = [x.clone() for x in (r if isinstance(r, (tuple, list)) else [r])]
    elif c["kind"] == "batches":
        embs = [rn(b, c["S"][i % 5], hid) for i, b in enumerate(c["sizes"])]
        r = eng.generate_many(embs, c["n"], eos_token_id=None, pad_token_id=2, coalesce=c["coalesce"], max_rows=c["max_rows"] if c["coalesce"] else None)
        for x in r:
            out += [y.clone() for y in (x if isinstance(x, (tuple, list)) else [x])]
    else:
        emb = rn(c["B"], c["S"], hid)
        o = um(inputs_embeds=emb, use_cache=True)
        tok = o.logits[:, -1].argmax(-1)
        step = um(input_ids=tok[:, None], past_key_values=o.past_key_values)
        out = [o.logits.clone(), step.logits.clone()]
    return out


for qwen in ((False, True) if __name__ == "__main__" else ()):
    model = build(qwen)
    um = model.base_model.model
    hid, V = um.config.hidden_size, um.lm_head.weight.shape[0]
    calls = make_calls(NCALL, hid, V)
    fresh = []
    for c in calls:
        um._engine.invalidate()
        try:
            fresh.append(run(model, c, hid, V))
        except Exception as e:      # noqa: BLE001
            fresh.append(e)
    um._engine.invalidate()
    n_same = 0
    for i, c in enumerate(calls):
        desc = f"{'qwen' if qwen else 'llama'} call {i}: {c}"
        try:
            got = run(model, c, hid, V)
        except Exception as e:      # noqa: BLE001
            if not isinstance(fresh[i], Exception) or type(fresh[i]) is not type(e):
                bad.append(desc + f" -> {type(e).__name__}: {str(e)[:200]} (fresh state: {'ok' if not isinstance(fresh[i], Exception) else type(fresh[i]).__name__})")
            continue
        if isinstance(fresh[i], Exception):
            bad.append(desc + f" -> ok with carried state, {type(fresh[i]).__name__}: {str(fresh[i])[:160]} with fresh state"); continue
        if len(got) != len(fresh[i]) or not all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(got, fresh[i])):
            which = [k for k, (a, b) in enumerate(zip(got, fresh[i])) if a.shape != b.shape or not torch.equal(a, b)]
            bad.append(desc + f" -> results {which} differ between carried and fresh engine state")
        else:
            n_same += 1
    torch.cuda.synchronize()
    print(f"{'qwen' if qwen else 'llama'}: {n_same} of {len(calls)} calls bit-identical between carried and fresh engine state; "
          f"state at the end: {len(um._engine._kv)} KV slots, {len(um._engine._dec)} decode states, {len(um._engine._ws)} workspaces", flush=True)
    del model, um
    torch.cuda.empty_cache()
if __name__ == "__main__":
    print(f"{2 * NCALL} calls computed, {len(bad)} failures")
    for b_ in bad[:40]: print("FAIL", b_)
    sys.exit(1 if bad else 0)
