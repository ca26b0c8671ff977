"""Parity bounds COMPUTED from the oracle instead of constants: for every component of the path scripts/parity_floor.rows()
runs the oracle three ways on the reference-recorded fixtures - fp32, the bf16-OPERAND FLOOR (only what a matrix instruction consumes rounded:
no bf16-MFMA implementation can be closer to fp32) and the bf16-STORAGE emulation (the HIP path's storage points, exact arithmetic between them)
- and a comparison of the HIP path with the fp32 reference is held to

        hip  <=  FACTOR x max(floor, storage emulation)         (FACTOR = 1.5: accumulation order and 1-ulp flips of rounded activations)

of ITS component.  Computed once per test session on the CPU (~10 s)."""
import functools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FACTOR = 1.5
# HIP vs the storage EMULATION (not vs fp32): both sit within ~max(floor, storage) of the fp32 result, on different sides of every rounding
# boundary they straddle, so their mutual distance is bounded by the sum of the two distances
FACTOR_VS_EMULATION = 2.5


@functools.lru_cache(maxsize=None)
def _rows():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    try:
        import parity_floor
    finally:
        sys.path.pop(0)
    import torch
    torch.manual_seed(0)
    return tuple(parity_floor.rows())


def rows(key: str):
    """All rows of the component whose name contains `key` (a multi-output component has one row per output)."""
    r = [x for x in _rows() if key in x["what"]]
    if not r:
        raise KeyError(f"no parity-floor row matches {key!r}; have {[x['what'] for x in _rows()]}")
    return r


def bound(key: str, factor: float = FACTOR) -> float:
    """factor x the largest max(floor, storage emulation) over the rows of the component `key`, relative to max |fp32 reference|."""
    return factor * max(max(r["floor"], r["storage_emulation"]) for r in rows(key))


def enc(fixture_hint: str = "") -> float:
    """Encoder-side comparisons that have no row of their own (e.g. spliced inputs_embeds of another batch shape of the same fixture)."""
    return bound(fixture_hint + ": inputs_embeds") if fixture_hint else max(bound(k) for k in ("clip_tiny", "beats_tiny", "Projector"))


def dec(fixture: str = "full_tiny_llama") -> float:
    """Decoder logits / hidden states of `fixture`'s stack under any batch regime: the larger of its prefill and per-step rows."""
    return max(bound(f"{fixture}: decoder prefill logits"), bound(f"{fixture}: end to end"))


def decoder_bound(emb, W, cfg, ref_ids, W_stored=None, factor: float = FACTOR) -> float:
    """For a decoder stack that has no fixture row (a test's own random model / embeddings): the oracle TEACHER-FORCED along `ref_ids` in fp32, as
    the operand floor and as the storage emulation (on W_stored = the parameters as the HIP modules hold them; default tests.util.stored_params(W))
    -> factor x max(floor, storage) of the per-step last-row logits, relative to max |fp32 logits|."""
    import torch
    from oracle import crab_oracle as O
    from tests.util import stored_params
    Ws = W_stored if W_stored is not None else stored_params(W)

    def run(Wx, e):
        cache = O.KVCache()
        logits, _, cache = O.decoder_forward(emb.float(), Wx, cfg, cache, last_only=True, emulate=e)
        out = [logits[:, -1]]
        for s_ in range(1, ref_ids.shape[1]):
            tok = Wx["model.embed_tokens.weight"].float()[ref_ids[:, s_ - 1]][:, None]
            logits, _, cache = O.decoder_forward(O._r(tok, e), Wx, cfg, cache, last_only=True, emulate=e)
            out.append(logits[:, -1])
        return torch.stack(out, 1)
    ref = run(W, None)
    sc = ref.abs().max().item()
    flo = (run(W, O.OPERANDS) - ref).abs().max().item() / sc
    sto = (run(Ws, torch.bfloat16) - ref).abs().max().item() / sc
    return factor * max(flo, sto)
