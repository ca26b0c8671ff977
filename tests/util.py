"""Shared helpers for the parity tests: fixture loading and weight regeneration."""
from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import numpy as np
import torch

from crab_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name: str) -> Tuple[dict, Dict[str, torch.Tensor]]:
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    arrs = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrs


def weights_from_table(meta: dict, verify: bool = True) -> Dict[str, torch.Tensor]:
    """Regenerate the synthetic checkpoint a fixture was made with; checks every checksum."""
    W = {}
    seed = meta["seed"]
    for name, shape, cks in meta["table"]:
        W[name] = synth.synth_tensor(name, shape, seed)
    # tied BEATs table: every layer aliases layer 0's relative_attention_bias (backbone.py:78-81)
    for name in list(W):
        if name.endswith("self_attn.relative_attention_bias.weight") and ".layers.0." not in name:
            pre = name.split(".layers.")[0]
            W[name] = W[pre + ".layers.0.self_attn.relative_attention_bias.weight"]
    if verify:
        for name, shape, cks in meta["table"]:
            got = synth.checksum(W[name])
            assert abs(got - cks) <= 1e-6 * max(1.0, abs(cks)), f"weight regeneration drifted for {name}"
    return W


def strip(W: Dict[str, torch.Tensor], prefix: str) -> Dict[str, torch.Tensor]:
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in W.items()}
