"""Differential fuzz through the C-ABI at random shapes around every dispatch boundary (scripts/fuzz_gemm.py, scripts/fuzz_attn.py): the GEMM
regimes x epilogues and the attention entry points x masks against fp32 torch on the same bf16 operands; and of the whole decoder path on random
tiny configurations (scripts/fuzz_decoder.py: Llama / Qwen2 shapes, GQA, adapter ranks, batch 1 .. 260, both layer sequencers, graph and eager) against
the fp32 CPU oracle, and of prepare_multimodal_inputs on random tiny encoder configurations (scripts/fuzz_multimodal.py: CLIP / BEATs / Q-Former
widths, depths, selected levels, frames, audio windows, ragged prompts) - both bounded by the oracle's own bf16-storage emulation on the same configuration; and of the RoPE / KV-append
fusions of the q|k|v projection against the unfused pair they replace, bit for bit (scripts/fuzz_rope_epilogue.py).  A combination outside a stated limit
must be REJECTED (CRAB_E_INVALID / CRAB_E_UNSUPPORTED), never computed wrong.  Fixed seeds: the cases are the same on every run."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,cases,seed", [("fuzz_gemm.py", 1200, 11), ("fuzz_attn.py", 400, 12), ("fuzz_decoder.py", 8, 13), ("fuzz_multimodal.py", 6, 14), ("fuzz_rope_epilogue.py", 120, 15), ("fuzz_frontend.py", 40, 16), ("fuzz_ops.py", 600, 17)])
def test_differential_fuzz(script, cases, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(cases), str(seed)], capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " 0 failures" in r.stdout, tail
