"""Import shims for driving the read-only reference tree (/root/reference) in THIS container.

Only used by tests/golden/make_golden.py to generate golden vectors.  Nothing here travels to the GPU
box as behaviour: the reference cannot be imported there, the committed .npz fixtures are the pins.
The list of shims follows SURVEY.md 8c (torchaudio stub, transformers 5.x moved helpers, the missing
models.video_llama2 module, hard-wired cluster paths).  No reference source is copied.
"""
from __future__ import annotations

import importlib.machinery
import os
import sys
import types

REF = os.environ.get("CRAB_REFERENCE", "/root/reference")


def install():
    if not os.path.isdir(REF):
        raise RuntimeError(f"reference tree not found at {REF}")
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)

    import transformers  # noqa: F401  (must be first)
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    # torchaudio is only reached from BEATs.preprocess (BEATs.py:119-132), never from extract_features
    def _stub(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        return m

    if "torchaudio" not in sys.modules:
        ta = _stub("torchaudio")
        tac = _stub("torchaudio.compliance")
        tak = _stub("torchaudio.compliance.kaldi")

        def _fbank(*a, **k):
            raise RuntimeError("torchaudio stub: fbank is outside the oracle's path")

        tak.fbank = _fbank
        tac.kaldi = tak
        ta.compliance = tac
        sys.modules.update({"torchaudio": ta, "torchaudio.compliance": tac, "torchaudio.compliance.kaldi": tak})

    for fn in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, fn) and hasattr(pu, fn):
            setattr(mu, fn, getattr(pu, fn))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        if hasattr(pu, "find_pruneable_heads_and_indices"):
            mu.find_pruneable_heads_and_indices = pu.find_pruneable_heads_and_indices
        else:
            mu.find_pruneable_heads_and_indices = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())

    # models/unified_arch.py:13 imports a module that is not in the tree
    if "models.video_llama2" not in sys.modules:
        pk = _stub("models.video_llama2")
        pk.__path__ = []
        pj = _stub("models.video_llama2.projector")
        pj.STCConnectorV35 = type("STCConnectorV35", (), {})
        pk.projector = pj
        sys.modules["models.video_llama2"] = pk
        sys.modules["models.video_llama2.projector"] = pj

    import models.Qformer as Q

    Q.BertPreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    Q.BertModel.get_head_mask = lambda self, hm, n, *a, **k: [None] * n
    if not hasattr(Q.BertModel, "invert_attention_mask") or True:
        import torch

        def _invert(self, m):
            if m.dim() == 3:
                e = m[:, None, :, :]
            else:
                e = m[:, None, None, :]
            e = e.to(dtype=self.dtype if hasattr(self, "dtype") else torch.float32)
            return (1.0 - e) * -10000.0

        Q.BertModel.invert_attention_mask = _invert
    return Q


def tiny_bert_config(hidden=128, heads=2, inter=256):
    from transformers.models.bert.configuration_bert import BertConfig

    return BertConfig(hidden_size=hidden, num_attention_heads=heads, intermediate_size=inter,
                      vocab_size=64, max_position_embeddings=64, hidden_dropout_prob=0.0,
                      attention_probs_dropout_prob=0.0)


def patch_bert_config(cfg_factory):
    """multimodal_encoder.py:105,206 call BertConfig.from_pretrained(<cluster path>)."""
    import models.multimodal_encoder as me

    class _Cfg:
        @staticmethod
        def from_pretrained(*a, **k):
            return cfg_factory()

    me.BertConfig = _Cfg
    return me
