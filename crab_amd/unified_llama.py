"""Mirror of reference `models/unified_llama.py`: `UnifiedConfig`, `UnifiedModel`, `UnifiedForCausalLM` with the same
`generate(batch_input_ids, batch_labels, batch_X_modals, batch_task_names, **kw)` / `forward(...)` surface
(unified_llama.py:26-391), on the MI355X HIP path.

Differences a caller can observe, all deliberate:
  * decoding is greedy unless the caller passes `do_sample=True` (the reference inherits sampling from the checkpoint's
    generation_config, A.7: Llama-2-chat ships temperature 0.6 / top_p 0.9, HF's default top_k 50 - these are the defaults of
    `do_sample=True` here; draws come from a counter-based generator keyed by `seed`, not from torch's global stream).
  * like the reference's generate(), `attention_mask` / `position_ids` from prepare_multimodal_inputs are NOT forwarded
    to the decoder (unified_llama.py:261-267): left pads are attended and positions run 0..S-1 (A.1) -- reproduced.
    The reference's forward() DOES pass them on (unified_llama.py:129-160, the training-time batch path) and so does
    forward() here: a LEFT-padded attention_mask becomes a per-sequence first visible key in the attention kernels and
    position_ids become explicit rotary positions (golden: forward_masked_tiny_llama.npz); any other 2-D mask (holes anywhere, which
    HF's mask utilities accept) becomes one visibility bit per key (golden: forward_holes_tiny_llama.npz).  The logits of a query
    that sees no key at all (a pad row) are undefined in the reference (implementation-dependent softmax over an all-masked row)
    and finite garbage here; no valid row depends on them.
  * the model lives in bf16 on the GPU (the whole-model bf16 conversion of inference_hyper_lora.py:1470, A.8).
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from . import ops
from .decoder import DecoderConfig, DecoderModel, GenerationEngine, LMHead
from .peft_hyper import PackedLinearGroup
from .unified_arch import UnifiedMetaForCausalLM, UnifiedMetaModel

BF16 = torch.bfloat16


class UnifiedConfig(DecoderConfig):
    model_type = "unified_llm"


class UnifiedModel(UnifiedMetaModel, DecoderModel):
    config_class = UnifiedConfig

    def __init__(self, config: DecoderConfig, device="cuda"):
        DecoderModel.__init__(self, config, device)
        self.config = config
        self.pad_token_id = config.pad_token_id if config.pad_token_id is not None else 0


class CausalLMOutput:
    def __init__(self, logits, hidden_states=None, past_key_values=None):
        self.logits, self.hidden_states, self.past_key_values, self.loss = logits, hidden_states, past_key_values, None


class UnifiedForCausalLM(nn.Module, UnifiedMetaForCausalLM):
    config_class = UnifiedConfig

    def __init__(self, config: DecoderConfig, device="cuda", **kwargs):
        nn.Module.__init__(self)
        self.config = config
        self.model = UnifiedModel(config, device=device)
        self.vocab_size = config.vocab_size
        self.lm_head = LMHead(config.hidden_size, config.vocab_size, device)
        self.is_avs_task = False
        self._engine = GenerationEngine(self.model, self.lm_head)
        self._past = None

    # ------------------------------------------------------------------ plumbing
    def get_model(self) -> UnifiedModel:
        return self.model

    def packed_groups(self) -> List[PackedLinearGroup]:
        return [g for layer in self.model.layers for g in layer.groups()]

    def _invalidate_graphs(self):
        self._engine.invalidate()

    def _apply(self, fn, *args, **kwargs):
        """.to(device) / .cuda() / .cpu(): nn.Module._apply maps every Parameter separately, which would turn the view
        Parameters of the packed projection groups into independent copies and leave the packed operands the GEMMs read
        behind.  Map the packed buffers themselves, re-point the views, drop captured graphs / caches.  dtype changes raise."""
        groups = self.packed_groups()
        packed = [(g.W, g.bias, g.RA, g.B2) for g in groups]
        # refuse a dtype change BEFORE anything is converted (rebind would only notice after nn.Module._apply had already mapped every
        # other parameter, leaving a half-converted model)
        if fn(torch.empty(1, dtype=BF16, device=self.lm_head.weight.device)).dtype != BF16:
            raise TypeError("crab_amd keeps decoder weights in bfloat16: .float() / .half() / .to(dtype) are not supported")
        # detach the view Parameters of the packed groups while nn.Module._apply maps the rest: mapping each view would
        # materialise a second copy of every projection matrix on the target device before rebind() replaces it
        views = []
        for g in groups:
            for lin in g.linears:
                for mod in lin.modules():                     # the Linear and its lora_route / lora_A / lora_B{i} holders
                    for name in list(mod._parameters):
                        views.append((mod, name, mod._parameters[name]))
                        mod._parameters[name] = None
        try:
            r = super()._apply(fn, *args, **kwargs)
        finally:
            for mod, name, prm in views:
                mod._parameters[name] = prm
        for g, (W, b, RA, B2) in zip(groups, packed):
            g.W, g.bias, g.RA, g.B2 = W, b, RA, B2             # the buffers as they were: rebind maps them once
            g.rebind(fn)
        self._invalidate_graphs()
        return r

    def get_input_embeddings(self):
        return self.model.embed_tokens

    def get_output_embeddings(self):
        return self.lm_head

    def resize_token_embeddings(self, new_num_tokens: int):
        """HF semantics: keep the old rows, new rows ~ N(0, 0.02) (initializer_range); both tables resized."""
        old = self.model.embed_tokens.weight
        if new_num_tokens == old.shape[0]:
            return self.model.embed_tokens
        dev, D = old.device, old.shape[1]
        n_keep = min(old.shape[0], new_num_tokens)
        for holder, attr in ((self.model.embed_tokens, "weight"), (self.lm_head, "weight")):
            w_old = getattr(holder, attr)
            g = torch.Generator(device="cpu").manual_seed(42)
            w_new = (0.02 * torch.randn((new_num_tokens, D), generator=g)).to(device=dev, dtype=BF16)
            w_new[:n_keep].copy_(w_old[:n_keep])
            setattr(holder, attr, nn.Parameter(w_new, requires_grad=False))
        self.model.embed_tokens.num_embeddings = new_num_tokens
        self.vocab_size = self.config.vocab_size = new_num_tokens
        self._invalidate_graphs()
        return self.model.embed_tokens

    @property
    def dtype(self):
        return BF16

    def eval(self):
        return super().eval()

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(
        self,
        batch_input_ids=None,
        batch_labels=None,
        batch_X_modals=None,
        batch_task_names=None,
        input_ids: torch.LongTensor = None,
        attention_mask: Optional[torch.Tensor] = None,
        position_ids: Optional[torch.LongTensor] = None,
        past_key_values=None,
        inputs_embeds: Optional[torch.Tensor] = None,
        labels: Optional[torch.LongTensor] = None,
        use_cache: Optional[bool] = None,
        output_attentions: Optional[bool] = None,
        output_hidden_states: Optional[bool] = None,
        return_dict: Optional[bool] = None,
        **kwargs,
    ):
        """Inference forward (unified_llama.py:47-161): the 1-token decode shortcut (:125-127), the multimodal
        branch (:129-146) and the plain inputs_embeds branch.  Returns an object with `.logits` (fp32, all rows),
        `.hidden_states` (tuple ending with the post-final-norm states when requested) and `.past_key_values`."""
        if labels is not None or self.is_avs_task:
            raise NotImplementedError("training losses / AVS forward are outside the inference hot path")
        eng = self._engine
        dev = self.device
        if input_ids is not None and input_ids.shape[1] == 1 and past_key_values is not None:
            kc, vc, n = past_key_values
            if n >= kc.shape[3]:
                # the position travels as a device word (graph-friendly), so the kernels cannot check it: an append past
                # Tmax would write into the next head's rows / past the allocation
                raise ValueError(f"KV cache is full: position {n} >= capacity {kc.shape[3]} (the prefill call sized it as "
                                 f"round64(S + 64)); re-run the prefill with a longer cache")
            B = input_ids.shape[0]
            kv_start, key_mask = self._key_visibility(attention_mask, B, n + 1, kc.shape[3])
            pos_ids = None
            if position_ids is not None and not bool((position_ids.reshape(-1) == n).all()):
                pos_ids = self._rotary_positions(position_ids, B, 1, kc.shape[3])
                eng._rope_tab(self._rope_need)
            emb = self.model.embed_tokens(input_ids.reshape(-1))
            ws = eng._workspace(B)
            ops.cast_rows(emb, ws.x, B, emb.shape[1])
            pos = torch.full((1,), n, device=dev, dtype=torch.int32)
            x, hfin = eng._layers(ws, B, 1, kc, vc, 0, kc.shape[3], 0, pos, None, pos_ids=pos_ids, kv_start=kv_start, key_mask=key_mask)
            hn = hfin.clone()
            logits = ops.gemm(hn, self.lm_head.weight, out_fp32=True)
            return CausalLMOutput(logits.view(B, 1, -1), (hn.view(B, 1, -1),) if output_hidden_states else None, (kc, vc, n + 1))
        if inputs_embeds is None and batch_input_ids is not None:
            # the multimodal branch (unified_llama.py:129-146): mask and positions of the padded batch go to the decoder
            inputs = self.prepare_multimodal_inputs(batch_input_ids=batch_input_ids, batch_labels=batch_labels,
                                                    batch_X_modals=batch_X_modals, batch_task_names=batch_task_names)
            inputs_embeds = inputs['inputs_embeds']
            attention_mask, position_ids = inputs['attention_mask'], inputs['position_ids']
        elif inputs_embeds is None and input_ids is not None:
            inputs_embeds = self.model.embed_tokens(input_ids)
        inputs_embeds = inputs_embeds.to(device=dev, dtype=BF16)
        B, S, _ = inputs_embeds.shape
        Tmax = (S + 64 + 63) // 64 * 64 if use_cache else (S + 63) // 64 * 64
        kv_start, key_mask = self._key_visibility(attention_mask, B, S, S)
        pos_ids = None
        if position_ids is not None and not bool((position_ids.reshape(-1, S).cpu() == torch.arange(S)).all()):
            pos_ids = self._rotary_positions(position_ids, B, S, Tmax)
            eng._rope_tab(self._rope_need)
        kc, vc = eng.alloc_cache(B, Tmax)
        logits, hn = eng.prefill(inputs_embeds, kc, vc, all_logits=True, pos_ids=pos_ids, kv_start=kv_start, key_mask=key_mask)
        return CausalLMOutput(logits, (hn,) if output_hidden_states else None, (kc, vc, S) if use_cache else None)

    def _key_visibility(self, attention_mask, B: int, T: int, width: int):
        """2-D attention_mask [B, T] over ALL keys (cached + new) -> (kv_start, key_mask) for the attention kernels, at most one of them set:
        nothing masked -> (None, None); left padding (zeros, then ones: what prepare_multimodal_inputs builds, unified_arch.py:344-348) ->
        int32 [B] index of the first visible key per sequence (whole key tiles below it are skipped); any other mask (holes anywhere: HF's
        mask utilities AND the padding mask with the causal one whatever its shape) -> int32 [B, ceil(width / 32)] visibility words
        (ops.pack_key_mask), `width` >= T bits wide (the cache capacity for the decode shortcut, whose context length travels as a device word)."""
        if attention_mask is None:
            return None, None
        m = attention_mask.to(torch.bool).reshape(B, -1)
        if m.shape[1] != T:
            raise ValueError(f"attention_mask covers {m.shape[1]} keys, expected {T} (cached + new tokens)")
        if bool(m.all()):
            return None, None
        start = (~m).sum(1)
        if bool((m == (torch.arange(T, device=m.device)[None] >= start[:, None])).all()):
            return start.to(device=self.device, dtype=torch.int32), None
        full = torch.zeros(B, max(width, T), dtype=torch.bool, device=m.device)
        full[:, :T] = m
        return None, ops.pack_key_mask(full).to(self.device)

    def _rotary_positions(self, position_ids, B: int, S: int, Tmax: int):
        """position_ids [B | 1, S] -> contiguous int32 [B, S] on the device (the caller grows the RoPE table to cover them)."""
        p = position_ids.reshape(-1, S)
        if p.shape[0] not in (1, B):
            raise ValueError(f"position_ids has {p.shape[0]} rows for a batch of {B}")
        lo, hi = int(p.min()), int(p.max())
        if lo < 0:
            raise ValueError("negative position_ids")
        self._rope_need = max(hi + 1, Tmax)
        return p.expand(B, S).to(device=self.device, dtype=torch.int32).contiguous()

    # ------------------------------------------------------------------ generate
    @torch.no_grad()
    def generate(self, batch_input_ids=None, batch_labels=None, batch_X_modals=None, batch_task_names=None, **kwargs):
        """unified_llama.py:244-267.  kwargs understood (HF names): max_new_tokens, min_new_tokens, eos_token_id,
        pad_token_id, use_cache, do_sample (+ temperature, top_k, top_p, seed), output_logits / return_dict_in_generate (parity audits),
        output_first_logits (ids + the fp32 logits of the first generated position, [B, V]: the record the multi-GPU eval gathers),
        inputs_embeds (skip prepare_multimodal_inputs)."""
        self._check_generate_kwargs(kwargs)
        sampling = self._sampling(kwargs)
        embeds = kwargs.pop("inputs_embeds", None)
        if ops.PROFILER is not None:
            ops.PROFILER.mark("encode_begin")
        if embeds is None:
            inputs = self.prepare_multimodal_inputs(batch_input_ids=batch_input_ids, batch_labels=batch_labels,
                                                    batch_X_modals=batch_X_modals, return_multi_scale_features=False,
                                                    return_gt_mask=False, batch_task_names=batch_task_names)
            embeds = inputs['inputs_embeds']
        embeds = embeds.to(device=self.device, dtype=BF16)
        max_new = int(kwargs.get("max_new_tokens", 20))
        eos = kwargs.get("eos_token_id", self.config.eos_token_id)
        pad = kwargs.get("pad_token_id", self.model.pad_token_id if self.model.pad_token_id is not None else eos)
        want_logits = bool(kwargs.get("output_logits")) and bool(kwargs.get("return_dict_in_generate"))
        want_first = bool(kwargs.get("output_first_logits"))      # not an HF name: ids + the first step's logits only (eval gather)
        res = self._engine.generate(embeds, max_new, eos_token_id=eos, pad_token_id=pad,
                                    min_new_tokens=int(kwargs.get("min_new_tokens", 0) or 0),
                                    prefill_chunk=int(kwargs.get("prefill_chunk", 0)), use_graph=kwargs.get("use_graph", True),
                                    return_step_logits=want_logits, decode_streams=int(kwargs.get("decode_streams", 1)),
                                    return_first_logits=want_first, sampling=sampling)
        if want_logits or want_first:
            res = list(res)
            out = type("GenerateOutput", (), {})()
            out.sequences = res.pop(0)
            if want_logits:
                sl = res.pop(0)
                out.logits = tuple(sl[:, i] for i in range(sl.shape[1]))
            if want_first:
                out.first_logits = res.pop(0)
            return out
        return res

    @torch.no_grad()
    def generate_batches(self, batches, coalesce: bool = False, max_rows: Optional[int] = None, **kwargs):
        """Throughput form of the eval loop (scripts/finetune/inference_hyper_lora.py:1466-1479 calls generate() once per collated batch of 8):
        `batches` = a list of dicts with the four generate() arguments (batch_input_ids, batch_labels, batch_X_modals, batch_task_names).  Every
        batch keeps its own prepare_multimodal_inputs result (its own left padding, like a separate call), then all of them decode together.
        coalesce = False: IN FLIGHT (GenerationEngine.generate_many: one decode group, graph and HIP stream per batch; ids equal to what
        generate() returns for each batch, bit for bit).
        coalesce = True: as ONE ragged decode batch (right-aligned in one KV cache, per-row rotary offset and first visible key): the weights
        stream once per step for all batches and the encoders run over the clips of all batches together, so batches of 8 reach the
        throughput of one large generate(); per-batch ids / logits agree with separate calls within the decoder's bf16 tolerance.
        Returns one id tensor per batch (with output_first_logits=True: (ids, fp32 logits of the first generated position))."""
        for k in ("output_logits", "return_dict_in_generate", "inputs_embeds"):
            if kwargs.get(k) is not None and kwargs.get(k) is not False:
                raise NotImplementedError(f"generate_batches returns token ids only: {k} is a generate() argument")
        want_first = bool(kwargs.get("output_first_logits"))
        self._check_generate_kwargs(kwargs)
        sampling = self._sampling(kwargs)
        if ops.PROFILER is not None:
            ops.PROFILER.mark("encode_begin")
        if coalesce:
            inputs = self.prepare_multimodal_inputs_many(batches, return_multi_scale_features=False, return_gt_mask=False)
            embeds = [d['inputs_embeds'].to(device=self.device, dtype=BF16) for d in inputs]
        else:
            embeds = []
            for b in batches:
                inputs = self.prepare_multimodal_inputs(batch_input_ids=b["batch_input_ids"], batch_labels=b.get("batch_labels"),
                                                        batch_X_modals=b["batch_X_modals"], return_multi_scale_features=False, return_gt_mask=False,
                                                        batch_task_names=b.get("batch_task_names"))
                embeds.append(inputs['inputs_embeds'].to(device=self.device, dtype=BF16))
        eos = kwargs.get("eos_token_id", self.config.eos_token_id)
        pad = kwargs.get("pad_token_id", self.model.pad_token_id if self.model.pad_token_id is not None else eos)
        return self._engine.generate_many(embeds, int(kwargs.get("max_new_tokens", 20)), eos_token_id=eos, pad_token_id=pad,
                                          min_new_tokens=int(kwargs.get("min_new_tokens", 0) or 0), use_graph=kwargs.get("use_graph", True),
                                          sampling=sampling, return_first_logits=want_first, coalesce=coalesce, max_rows=max_rows)

    # HF generate() arguments that would CHANGE what is decoded and that this path does not implement: refused by name instead of ignored
    # (name -> the value that means "off").  Everything the reference's loops pass (use_cache, max_new_tokens; do_sample & co. from the
    # checkpoint's generation_config) is implemented; unknown names that cannot alter the ids (output_attentions = False, ...) pass through.
    _UNSUPPORTED = {"num_beams": 1, "num_beam_groups": 1, "num_return_sequences": 1, "repetition_penalty": 1.0, "no_repeat_ngram_size": 0,
                    "encoder_repetition_penalty": 1.0, "length_penalty": 1.0, "penalty_alpha": None, "bad_words_ids": None, "force_words_ids": None,
                    "logits_processor": None, "stopping_criteria": None, "prefix_allowed_tokens_fn": None, "constraints": None, "typical_p": 1.0,
                    "epsilon_cutoff": 0.0, "eta_cutoff": 0.0, "diversity_penalty": 0.0, "suppress_tokens": None, "begin_suppress_tokens": None,
                    "forced_bos_token_id": None, "forced_eos_token_id": None, "assistant_model": None, "streamer": None, "max_time": None,
                    "stop_strings": None, "min_p": None, "guidance_scale": None, "sequence_bias": None}

    @classmethod
    def _check_generate_kwargs(cls, kwargs):
        for k, off in cls._UNSUPPORTED.items():
            v = kwargs.get(k, off)
            if v is None or v == off or (isinstance(v, (list, tuple)) and len(v) == 0):
                continue
            raise NotImplementedError(f"generate({k}={v!r}): not implemented on this path (greedy and HF sample mode with temperature / top_k / top_p are); "
                                      "it would change the decoded ids, so it is refused rather than ignored")
        if "max_length" in kwargs and kwargs.get("max_length") is not None and kwargs.get("max_new_tokens") is None:
            raise NotImplementedError("generate(max_length=...): with inputs_embeds only HF counts new tokens alone - pass max_new_tokens")

    @staticmethod
    def _sampling(kwargs):
        """HF sample-mode arguments -> (temperature, top_k, top_p, seed) or None (greedy).  Defaults = what a Llama-2-chat checkpoint's
        generation_config + GenerationConfig's own defaults give the reference's generate() call (temperature 0.6, top_p 0.9, top_k 50)."""
        if not kwargs.get("do_sample"):
            return None
        t, k, p_ = float(kwargs.get("temperature", 0.6)), int(kwargs.get("top_k", 50) or 0), float(kwargs.get("top_p", 0.9))
        if t <= 0 or not (0 < p_ <= 1) or k < 0:
            raise ValueError("do_sample: temperature > 0, 0 < top_p <= 1, top_k >= 0")
        seed = kwargs.get("seed")
        if seed is None:
            # no explicit seed: like HF (whose torch.multinomial advances the global generator) every call draws a fresh stream - the seed is
            # taken from torch's default CPU generator, so torch.manual_seed(n) still reproduces a whole run while two calls on the same prompt
            # no longer return the same sample.  seed=<int> stays fully deterministic per call.  (The seed is baked into the captured decode
            # graph: an unseeded sampling call re-captures it.)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return (t, k, p_, int(seed) & 0x7FFFFFFFFFFFFFFF)

    @torch.no_grad()
    def generate_avs(self, batch_input_ids, batch_labels, batch_X_modals, batch_task_names, **kwargs):
        """unified_llama.py:270-361: greedy generation with the post-final-norm hidden state of every step kept; the states
        of the steps j with output_ids[0, j+1] in {<mask_0..5>} (bs == 1 assumed, :338) become the 6 prompt embeddings
        of the SegModule.  Returns {'output_ids', 'pred_masks'} (only 'output_ids' when != 6 mask tokens were produced).
        One deviation: step 0 contributes its LAST-row state (the reference's step-0 entry holds all S prompt rows).

        bs > 1 (r06): the reference's own bs > 1 behaviour is unusable (it reads row 0's mask positions for every row, :338, and its pixel loops
        call it with one sample, scripts/quick_start.py:270-450), so a batch here means `bs` INDEPENDENT bs-1 calls executed together
        (generate_avs_many): every sample keeps the prompt, positions and picks of its own call.  'output_ids' is then [bs, n_max] (rows that
        stopped earlier padded with pad_token_id, as HF pads finished rows) and 'pred_masks' a list with None for the rows that did not produce
        six mask tokens (the reference prints and returns ids only for such a sample, :345-352)."""
        if len(batch_input_ids) > 1:
            samples = [{"batch_input_ids": [batch_input_ids[i]], "batch_labels": [batch_labels[i]] if batch_labels is not None else None,
                        "batch_X_modals": [batch_X_modals[i]], "batch_task_names": [batch_task_names[i]]} for i in range(len(batch_input_ids))]
            res = self.generate_avs_many(samples, **kwargs)
            eos = kwargs.get("eos_token_id", self.config.eos_token_id)
            pad = kwargs.get("pad_token_id", self.model.pad_token_id if self.model.pad_token_id is not None else eos)
            n_max = max(r['output_ids'].shape[1] for r in res)
            ids = torch.full((len(res), n_max), int(pad if pad is not None else 0), device=res[0]['output_ids'].device, dtype=res[0]['output_ids'].dtype)
            for i, r in enumerate(res):
                ids[i, :r['output_ids'].shape[1]] = r['output_ids'][0]
            return {'output_ids': ids, 'pred_masks': [r['pred_masks'][0] if r.get('pred_masks') is not None else None for r in res]}
        return self.generate_avs_many([{"batch_input_ids": batch_input_ids, "batch_labels": batch_labels, "batch_X_modals": batch_X_modals,
                                        "batch_task_names": batch_task_names}], **kwargs)[0]

    @torch.no_grad()
    def generate_avs_many(self, samples, max_rows: Optional[int] = None, **kwargs):
        """The pixel-task loops as a THROUGHPUT path (BASELINE configs[4]): `samples` = a list of dicts with the four generate_avs() arguments, each
        one call of the reference's loops (scripts/quick_start.py:270-450: one sample per call).  Every call keeps its own
        prepare_multimodal_inputs result, prompt length and positions; all of them decode as ONE ragged batch (GenerationEngine.generate_many
        (coalesce=True, return_hidden=True): the weights stream once per step for every sample, the encoders see all images / audio windows
        together), the <mask_i> picks are made PER ROW (:333-352), and the rows that produced six mask tokens go through the SegModule together
        (crab_amd/seg_module.py: samples of one class count batched).  Returns one dict per call, exactly what generate_avs returns for it:
        {'output_ids', 'pred_masks'} - or {'output_ids'} alone when the call's row produced != 6 mask tokens (with the reference's message)."""
        inputs, outs = self._avs_generate(samples, max_rows, kwargs)
        seg_ids = {self.SPECIAL_TOKEN_2_IDS[f'<mask_{i}>'] for i in range(6)}
        chosen = []                                           # (call, row, the six step indices)
        n_tok = [ids.shape[1] for ids, _ in outs]
        flat = torch.cat([ids.reshape(-1) for ids, _ in outs]).tolist()      # one device -> host transfer for every call's ids
        o = 0
        for g, (ids, _) in enumerate(outs):
            for r in range(ids.shape[0]):
                row = flat[o:o + n_tok[g]]
                o += n_tok[g]
                picks = [j for j in range(len(row) - 1) if row[j + 1] in seg_ids]
                if len(picks) == 0:
                    print('len(pred_embeddings) == 0')
                elif len(picks) < 6:
                    print(f'pred_embeddings.shape[1] < 6, shape: {len(picks)}')
                else:
                    if len(picks) > 6:
                        print(f'pred_embeddings.shape[1] > 6, shape: {len(picks)}')
                    chosen.append((g, r, picks[-6:]))
        return self._avs_segment(samples, inputs, outs, chosen)

    def _avs_generate(self, samples, max_rows, kwargs):
        """First half of generate_avs_many: inputs of every call (multi-scale CLIP features kept) and the ragged decode with per-step hidden states."""
        self._check_generate_kwargs(kwargs)
        sampling = self._sampling(kwargs)
        if ops.PROFILER is not None:
            ops.PROFILER.mark("encode_begin")
        inputs = self.prepare_multimodal_inputs_many(samples, return_multi_scale_features=True, return_gt_mask=True)
        embeds = [d['inputs_embeds'].to(device=self.device, dtype=BF16) for d in inputs]
        eos = kwargs.get("eos_token_id", self.config.eos_token_id)
        pad = kwargs.get("pad_token_id", self.model.pad_token_id if self.model.pad_token_id is not None else eos)
        outs = self._engine.generate_many(embeds, int(kwargs.get("max_new_tokens", 20)), eos_token_id=eos, pad_token_id=pad,
                                          min_new_tokens=int(kwargs.get("min_new_tokens", 0) or 0), use_graph=kwargs.get("use_graph", True),
                                          sampling=sampling, coalesce=True, max_rows=max_rows, return_hidden=True)
        return inputs, outs

    def _avs_segment(self, samples, inputs, outs, chosen):
        """Second half: `chosen` = [(call, row, six step indices)] -> the picked states of those rows through the SegModule (batched per class
        count), one result dict per call."""
        results = [{'output_ids': ids} for ids, _ in outs]
        if not chosen:
            return results
        pred_embeddings = torch.stack([torch.stack([outs[g][1][r, j] for j in picks]) for g, r, picks in chosen])          # [n, 6, D]
        ms = [torch.stack([inputs[g]['multi_scale_image_features'][lv][r] for g, r, _ in chosen]) for lv in range(len(inputs[0]['multi_scale_image_features']))]
        tasks = [samples[g]["batch_task_names"][r] for g, r, _ in chosen]
        seg = self.model.postprocess_seg(pred_embeddings=pred_embeddings, multi_scale_image_feature_list=ms, gt_mask=None, batch_task_names=tasks)
        for (g, r, _), m in zip(chosen, seg['pred_masks']):
            bs_g = outs[g][0].shape[0]
            results[g].setdefault('pred_masks', [None] * bs_g)[r] = m
        return results

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Checkpoints are fp32 (reference ships --bf16 False); tensors are cast to the resident bf16 storage."""
        r = super().load_state_dict(state_dict, strict=strict, assign=False)      # copy_ casts to each parameter's dtype
        self._invalidate_graphs()
        return r
