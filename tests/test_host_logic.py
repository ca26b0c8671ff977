"""Host-side logic that needs no GPU: packed projection groups stay consistent under nn.Module._apply, forward() refuses
what the HIP path does not implement, results of generate() do not alias engine state."""
import pytest
import torch
from crab_amd import ops

from crab_amd.peft_hyper import LoraConfig, PackedLinearGroup, get_peft_model
from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM


def _tiny(device="cpu"):
    cfg = UnifiedConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, vocab_size=96, pad_token_id=2)
    return get_peft_model(UnifiedForCausalLM(cfg, device=device), LoraConfig())


def test_packed_group_views_alias_the_packed_buffers():
    g = PackedLinearGroup(["gate_proj", "up_proj"], 16, [8, 8], False, "cpu", interleave=True)
    g.attach_lora(8, 16, 3)
    g.linears[1].weight.data.fill_(3.0)
    assert float(g.W[1::2].min()) == 3.0 and float(g.W[0::2].abs().max()) == 0.0
    g.linears[0].lora_B1.weight.data.fill_(2.0)
    assert float(g.B2[0::2, 8:16].min()) == 2.0 and float(g.B2[1::2].abs().max()) == 0.0


def test_module_apply_keeps_views_bound_and_refuses_dtype_changes():
    model = _tiny()
    um = model.base_model.model
    g = um.model.layers[0].self_attn._qkv
    g.W.normal_()
    before = g.W.clone()
    model.to("cpu")                                              # nn.Module._apply with an identity map
    g2 = um.model.layers[0].self_attn._qkv
    assert torch.equal(g2.W, before)
    q = um.model.layers[0].self_attn.q_proj
    assert q.weight.data_ptr() == g2.W.data_ptr()                # still a view: a later load_state_dict fills the packed operand
    q.weight.data.zero_()
    assert float(g2.W[: q.out_features].abs().max()) == 0.0
    assert q.lora_A.weight.data_ptr() == g2.RA[g2.nl:].data_ptr()
    with pytest.raises(TypeError):
        model.float()
    # the refusal happens BEFORE anything is converted: every parameter keeps its storage (bf16; the norm weights fp32 since r05), the views
    # still alias the packed buffers
    from crab_amd import ops
    for name, p in model.named_parameters():
        want = ops.RMS_DTYPE if (p.dim() == 1 and "norm" in name) else torch.bfloat16
        assert p.dtype == want, (name, p.dtype)
    q = um.model.layers[0].self_attn.q_proj
    assert q.weight.data_ptr() == um.model.layers[0].self_attn._qkv.W.data_ptr()


def test_state_dict_round_trip_through_views():
    a, b = _tiny(), _tiny()
    for p in a.parameters():
        p.data.normal_()
    r = b.load_state_dict(a.state_dict(), strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    ga, gb = a.base_model.model.model.layers[1].mlp._gu, b.base_model.model.model.layers[1].mlp._gu
    assert torch.equal(ga.W, gb.W) and torch.equal(ga.RA, gb.RA) and torch.equal(ga.B2, gb.B2)


def test_forward_mask_handling_and_full_cache():
    um = _tiny().base_model.model
    emb = torch.zeros(2, 5, 64)
    mask = torch.ones(2, 5, dtype=torch.long)
    mask[1, :2] = 0                                              # left padding: a first visible key per sequence
    ks, km = um._key_visibility(mask, 2, 5, 5)
    assert ks.tolist() == [0, 2] and km is None and um._key_visibility(torch.ones(2, 5), 2, 5, 5) == (None, None)
    with pytest.raises(ValueError, match="covers 5 keys"):
        um._key_visibility(mask, 2, 6, 6)
    mask[1, 3] = 0                                               # a hole behind the padding: one visibility bit per key, `width` bits wide
    ks, km = um._key_visibility(mask, 2, 5, 64)
    assert ks is None and km.dtype == torch.int32 and km.shape == (2, 2) and km.tolist() == [[0b11111, 0], [0b10100, 0]]
    big = torch.ones(1, 70, dtype=torch.long); big[0, 31] = 0; big[0, 64] = 0
    assert ops.pack_key_mask(big).tolist() == [[2 ** 31 - 1, -1, 0b111110]]        # words above 2^31 - 1 wrap to signed int32
    assert um._rotary_positions(torch.tensor([[0, 0, 0, 1, 2]]), 2, 5, 64).tolist() == [[0, 0, 0, 1, 2]] * 2
    kc = torch.zeros(2, 1, 4, 64, 16)
    with pytest.raises(ValueError, match="KV cache is full"):
        um(input_ids=torch.zeros(1, 1, dtype=torch.long), past_key_values=(kc, kc.clone(), 64))


def test_plan_batch_splits_when_the_kv_cache_would_not_fit():
    """Capacity planning (scripts/quick_start.py:36-41 runs max_new_tokens = 500): the engine sizes a generate() from the bytes a
    sequence costs and the memory the device still has, and splits a batch that would not fit into even groups."""
    um = _tiny().base_model.model
    eng = um._engine
    per = eng.bytes_per_sequence(702, 500)
    c = um.config
    assert per >= 2 * c.num_hidden_layers * c.num_key_value_heads * 1216 * c.head_dim * 2           # at least the KV rows (Tmax = round64(1202))
    eng.kv_budget_bytes = int((eng.fixed_bytes(10, 702) + 3.5 * per) / 0.94) + 1
    assert eng.plan_batch(10, 702, 500) == [3, 3, 2, 2] and eng.last_plan["bytes_per_seq"] == per
    assert eng.plan_batch(3, 702, 500) == [3]
    eng.kv_budget_bytes = 1                                       # nothing fits: one sequence at a time, never zero
    assert eng.plan_batch(4, 702, 500) == [1, 1, 1, 1]
    # Llama-2-7B geometry: the reference's 500 new tokens at S = 766 cost 0.63 GiB of KV per clip -> 384 clips (240 GiB) cannot share
    # a 288 GB device with 14 GB of weights, 256 (160 GiB) can
    from crab_amd.decoder import DecoderConfig, DecoderModel, GenerationEngine
    big = GenerationEngine.__new__(GenerationEngine)
    big.cfg, big.lm_head, big._ws, big._kv, big.last_plan = DecoderConfig(), type("H", (), {"weight": torch.empty(32017, 1)})(), {}, {}, None
    big.kv_budget_bytes = 270 * 2 ** 30 - 15 * 2 ** 30
    assert big.plan_batch(256, 766, 500) == [256]
    assert big.plan_batch(384, 766, 500) == [192, 192]


def test_memory_budget_is_a_pure_query_and_stale_slots_are_evicted_lazily(monkeypatch):
    """After generate_many (slots 0 .. G-1) a single-group generate() counts EVERY persistent KV buffer of the engine as reclaimable - its own
    slot's (alloc_cache replaces it) and the other slots' (alloc_cache evicts them the moment the new cache does not fit beside them) - but
    memory_budget() itself frees nothing (a planning query used to drop live slots, graphs included, and a loop alternating
    generate_many(G > 1) with generate() re-allocated tens of GB per alternation).  r03's failure mode - stale slots counted but never
    freed, so the plan over-committed and ran out of memory - is closed by _evict_for."""
    from crab_amd.decoder import DecoderConfig, GenerationEngine
    eng = GenerationEngine.__new__(GenerationEngine)
    eng.cfg, eng.lm_head, eng.last_plan, eng.kv_budget_bytes = DecoderConfig(), type("H", (), {"weight": torch.empty(8, 1)})(), None, None
    kv = lambda n: (torch.empty(n, dtype=torch.bfloat16), torch.empty(n, dtype=torch.bfloat16))
    eng._kv = {(0,): kv(1000), (1,): kv(3000), (2,): kv(5000)}
    eng._dec = {0: object(), 1: object(), 2: object()}
    eng._ws = {"prefill": object(), ("decode", 8, 0): object(), ("decode", 8, 1): object(), ("decode", 8, 2): object()}
    emptied = []
    monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (10_000, 100_000))
    monkeypatch.setattr(torch.cuda, "memory_reserved", lambda dev=None: 700)
    monkeypatch.setattr(torch.cuda, "memory_allocated", lambda dev=None: 200)
    monkeypatch.setattr(torch.cuda, "empty_cache", lambda: emptied.append(1))
    every = 2 * 2 * (1000 + 3000 + 5000)
    assert eng.memory_budget(0, 0, slots=2) == 10_000 + 500 + every and eng._live_slots == 2
    assert eng.memory_budget(0, 0) == 10_000 + 500 + every and eng._live_slots == 1
    assert sorted(eng._kv) == [(0,), (1,), (2,)] and sorted(eng._dec) == [0, 1, 2] and len(eng._ws) == 4 and not emptied      # nothing freed
    # a new cache for slot 0 that fits beside the stale slots: nothing is evicted
    assert eng._evict_for(9_000, 0, slack=1_000) is False and sorted(eng._kv) == [(0,), (1,), (2,)] and not emptied
    # one that does not: the slots the running call does not use (>= _live_slots) go, with their decode states, graphs and workspaces
    assert eng._evict_for(10_000, 0, slack=1_000) is True
    assert sorted(eng._kv) == [(0,)] and sorted(eng._dec) == [0] and [k for k in eng._ws if k != "prefill"] == [("decode", 8, 0)] and emptied
    # a call planned over two slots keeps both while it allocates its second cache
    eng._kv = {(0,): kv(1000), (1,): kv(3000), (2,): kv(5000)}
    eng._dec = {0: object(), 1: object(), 2: object()}
    eng.memory_budget(0, 0, slots=2)
    assert eng._evict_for(50_000, 1) is True and sorted(eng._kv) == [(0,), (1,)]
    eng.kv_budget_bytes = 123                                     # the override never touches the device or the slots
    assert eng.memory_budget(0, 0) == 123


def test_generate_batches_refuses_generate_only_arguments():
    """generate_batches returns ids (+ the first-step logits with output_first_logits); the other per-call extras of generate() are refused
    loudly instead of being dropped."""
    from crab_amd.unified_llama import UnifiedForCausalLM
    class _Stub:
        _sampling = staticmethod(UnifiedForCausalLM._sampling)
    for k in ("output_logits", "return_dict_in_generate", "inputs_embeds"):
        with pytest.raises(NotImplementedError, match=k):
            UnifiedForCausalLM.generate_batches.__wrapped__(_Stub(), [], **{k: True})


def test_encoder_chunks_fill_whole_tile_rounds():
    """plan_enc_chunks: every partition covers n with calls of at most cmax blocks, and for the benchmark's 448 clips x 8 frames x 257 rows it
    avoids the 64-clip chunk (514 row tiles: 8.03 rounds of 256 tiles on the width-1024 projections = 9)."""
    from crab_amd.unified_arch import plan_enc_chunks

    def rounds(m, rows):
        rt = -(-m * rows // 256)
        return sum(-(-rt * ct // 256) * ku for ct, ku in ((12, 1), (4, 1), (16, 1), (4, 4)))
    for rows in (8 * 257, 10 * 257, 2 * 50, 0):
        for n in (1, 2, 5, 47, 64, 95, 96, 97, 200, 448, 1000):
            for cmax in (2, 64, 96):
                ch = plan_enc_chunks(n, rows, cmax)
                assert sum(ch) == n and all(0 < c <= cmax for c in ch), (n, rows, cmax, ch)
                if rows:
                    fixed = [cmax] * (n // cmax) + ([n % cmax] if n % cmax else [])
                    assert sum(rounds(c, rows) for c in ch) <= sum(rounds(c, rows) for c in fixed), (n, rows, cmax, ch)
    assert plan_enc_chunks(448, 8 * 257, 96) == [95, 95, 95, 95, 68]
    assert plan_enc_chunks(448, 8 * 257, 64) == [63] * 7 + [7]
    assert plan_enc_chunks(0, 100, 96) == [] and plan_enc_chunks(5, 100, 2) == [2, 2, 1]


def test_generate_refuses_hf_arguments_it_would_otherwise_ignore():
    """A drop-in must not silently decode something else: HF arguments that change the ids and are not implemented (beam search, penalties, processors,
    stopping criteria, max_length) raise by name; their neutral values and everything the reference's loops pass go through."""
    import pytest
    from crab_amd.unified_llama import UnifiedForCausalLM as U
    U._check_generate_kwargs(dict(max_new_tokens=500, use_cache=True, do_sample=True, temperature=0.6, top_k=50, top_p=0.9, num_beams=1,
                                  repetition_penalty=1.0, bad_words_ids=[], logits_processor=None, output_logits=True, return_dict_in_generate=True,
                                  max_length=20))                                      # max_length next to max_new_tokens: HF lets max_new_tokens win
    for kw in (dict(num_beams=4), dict(repetition_penalty=1.2), dict(no_repeat_ngram_size=3), dict(stopping_criteria=[object()]), dict(penalty_alpha=0.6),
               dict(num_return_sequences=2), dict(max_length=64), dict(typical_p=0.9)):
        with pytest.raises(NotImplementedError, match=list(kw)[0]):
            U._check_generate_kwargs(kw)
