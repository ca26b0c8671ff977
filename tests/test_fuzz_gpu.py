"""Differential fuzz through the C-ABI at random shapes around every dispatch boundary - short fixed-seed runs of the eleven generators under scripts/
(one process: scripts/fuzz_all.py; the cases are the same on every run):
  fuzz_gemm.py           the GEMM regimes x epilogues against fp32 torch on the same bf16 operands,
  fuzz_attn.py           the attention entry points x masks,
  fuzz_decoder.py        the whole decoder path on random tiny Llama / Qwen2 configurations (GQA, adapter ranks, batch 1 .. 260, both layer sequencers,
                         graph and eager; the EOS / min_new_tokens state machine; forward() under random 2-D masks) against the fp32 CPU oracle,
  fuzz_multimodal.py     prepare_multimodal_inputs on random tiny encoder stacks against the oracle - both bounded by the oracle's own bf16-storage
                         emulation on the same configuration,
  fuzz_rope_epilogue.py  the RoPE / KV-append fusions of the q|k|v projection against the unfused pair, bit for bit,
  fuzz_frontend.py       CLIP frame preprocessing and the kaldi fbank on random sizes / lengths against the numpy restatements,
  fuzz_ops.py            norms, embedding, casts, copies, the stand-alone router, SwiGLU, arg-max against torch; r06: the fp32 quantiser, the split-operand
                         GEMM, GroupNorm in its four forms,
  fuzz_seg.py            r06: the SegModule helpers, the remaining VQGAN helpers and the eval loops' metrics (counts equal to oracle/metrics_oracle.py),
  fuzz_engine_state.py   r06: random SEQUENCES of generate / generate_many / forward calls on one engine, carried state against fresh state, bit for bit,
  fuzz_model_state.py    r06: the same at the model level (encoders, projectors, splice, engine; changing frames / windows / modality subsets; a side stream),
  fuzz_vqgan.py          r06: the VQGAN mask tokenizer at random batch and mask sizes against oracle/vqgan_oracle.py (ids equal outside the oracle's own margin).
(r05: half the r04 case count here - the suite has a time limit; tests/test_slow_gpu.py runs 8x that (scale 4) under `-m gpu_slow`.)
A combination outside a stated limit must be REJECTED (CRAB_E_INVALID / CRAB_E_UNSUPPORTED), never computed wrong; outputs sit inside sentinel guards."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FUZZERS = ["fuzz_gemm.py", "fuzz_attn.py", "fuzz_decoder.py", "fuzz_multimodal.py", "fuzz_rope_epilogue.py", "fuzz_frontend.py", "fuzz_ops.py", "fuzz_seg.py", "fuzz_engine_state.py", "fuzz_model_state.py", "fuzz_vqgan.py"]


def test_differential_fuzz():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fuzz_all.py"), "0.5"], capture_output=True, text=True, timeout=1200)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    results = {l.split()[1]: l for l in r.stdout.splitlines() if l.startswith("RESULT ")}
    for f in FUZZERS:
        assert f in results and " rc=0 " in results[f] and " 0 failures" in results[f], (f, results.get(f), tail)
