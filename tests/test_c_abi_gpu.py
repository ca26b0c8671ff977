"""The C ABI driven from C: examples/decode_demo.c (gcc, libcrab_hip.so + the HIP runtime, no Python / torch in the process) runs prefill +
greedy decode of a tiny hyper-LoRA decoder from the packed weights written here, eagerly and through a HIP graph it captures itself;
its ids and per-step logits must equal the Python engine's bit for bit (same launches in the same order)."""
import os
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_demo(tmp_path):
    exe = str(tmp_path / "decode_demo")
    cmd = ["gcc", "-O1", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           os.path.join(ROOT, "examples", "decode_demo.c"), "-o", exe, "-L" + os.path.join(ROOT, "crab_amd"), "-lcrab_hip", "-L/opt/rocm/lib",
           "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "crab_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def _raw(t: torch.Tensor) -> bytes:
    return t.detach().contiguous().cpu().view(torch.int16).numpy().tobytes()


@pytest.mark.parametrize("qwen", [False, True])
def test_c_caller_matches_python_engine(tmp_path, qwen):
    from crab_amd.peft_hyper import LoraConfig, get_peft_model
    from crab_amd.unified_llama import UnifiedConfig, UnifiedForCausalLM
    torch.manual_seed(11 + qwen)
    D, I, L, H, Hk, V = 128, 352, 3, 2, (1 if qwen else 2), 320
    cfg = UnifiedConfig(hidden_size=D, intermediate_size=I, num_hidden_layers=L, num_attention_heads=H, num_key_value_heads=Hk, vocab_size=V,
                        attention_bias=qwen, pad_token_id=2)
    um = UnifiedForCausalLM(cfg, device="cuda")
    model = get_peft_model(um, LoraConfig())
    for p in model.parameters():
        p.data.copy_((torch.randn(p.shape) * 0.06).to(BF) if p.dim() > 1 else (1 + 0.1 * torch.randn(p.shape)).to(BF))
    lc = LoraConfig()
    B, S, n_new = 3, 11, 6
    emb = (torch.randn(B, S, D) * 0.7).to(BF).cuda()
    blob = [_raw(emb)]
    for layer in um.model.layers:
        for g in layer.groups():
            blob.append(_raw(g.W))
            if g.bias is not None:
                blob.append(_raw(g.bias))
            blob += [_raw(g.RA), _raw(g.B2)]
        blob += [_raw(layer.input_layernorm.weight), _raw(layer.post_attention_layernorm.weight)]
    blob += [_raw(um.model.norm.weight), _raw(um.lm_head.weight), _raw(um.model.embed_tokens.weight)]
    bpath, exe = str(tmp_path / "blob.bin"), _build_demo(tmp_path)
    with open(bpath, "wb") as f:
        f.write(b"".join(blob))
    ref_ids, ref_logits = um._engine.generate(emb, n_new, eos_token_id=None, pad_token_id=2, return_step_logits=True, use_graph=False)
    ref_ids, ref_logits = ref_ids.cpu().numpy(), ref_logits.float().cpu().numpy()          # [B, n], [B, n, V]
    for use_graph in (0, 1):
        opath = str(tmp_path / f"out{use_graph}.bin")
        args = [exe, bpath, opath] + [str(v) for v in (D, I, L, H, Hk, V, lc.lora_nums, lc.r, B, S, n_new, int(qwen), use_graph)]
        r = subprocess.run(args, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, (r.stdout, r.stderr)
        raw = open(opath, "rb").read()
        ids = np.frombuffer(raw[:B * n_new * 8], dtype=np.int64).reshape(B, n_new)
        logits = np.frombuffer(raw[B * n_new * 8:], dtype=np.float32).reshape(n_new, B, V).transpose(1, 0, 2)
        assert np.array_equal(ids, ref_ids), (use_graph, ids, ref_ids)
        assert np.array_equal(logits, ref_logits), (use_graph, np.abs(logits - ref_logits).max())
    assert len(set(ref_ids.reshape(-1).tolist())) > 1                    # not a degenerate constant output
