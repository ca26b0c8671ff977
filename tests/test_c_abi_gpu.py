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


def _build(tmp_path, name, extra=()):
    exe = str(tmp_path / name)
    cmd = ["gcc", "-O1", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           os.path.join(ROOT, "examples", name + ".c"), "-o", exe, "-L" + os.path.join(ROOT, "crab_amd"), "-lcrab_hip", "-L/opt/rocm/lib",
           "-lamdhip64", "-lm", "-Wl,-rpath," + os.path.join(ROOT, "crab_amd"), "-Wl,-rpath,/opt/rocm/lib"] + list(extra)
    subprocess.check_call(cmd)
    return exe


def test_c_caller_runs_a_whole_clip(tmp_path):
    """examples/clip_demo.c: CLIP tower -> VLProjector, BEATs -> ALProjector (crab_clip_layer / crab_beats_layer / crab_qformer_layer), the
    splice of prepare_multimodal_inputs, decoder prefill + greedy decode (crab_llama_layers), all from C on the weights of the tiny Crab of
    the reference-recorded fixture: its inputs_embeds, ids and per-step logits equal the Python modules' bit for bit, and its ids equal
    the ids the REFERENCE generated for this clip (full_tiny_llama.npz)."""
    from crab_amd import synth
    from tests.util import build_tiny_crab, load_fixture, weights_from_table
    meta, A = load_fixture("full_tiny_llama")
    model = build_tiny_crab(meta)
    model.load_state_dict(weights_from_table(meta), strict=False)
    um = model.base_model.model
    inner = um.model
    p = meta["prompts"]
    ids0 = A["ids0"]
    mods = {'<video>': synth.synth_video(p["t_v"], seed=meta["seed"], clip=p["clip0"]),
            '<audio>': synth.synth_audio(p["t_a"], p["l_a"], seed=meta["seed"], clip=p["clip0"])}
    n_new = meta["new_tokens"]
    inp = model.prepare_multimodal_inputs([ids0], [torch.full_like(ids0, -100)], [mods], ['avqa'])
    emb_ref = inp["inputs_embeds"]
    ref_ids, ref_logits = um._engine.generate(emb_ref, n_new, eos_token_id=None, pad_token_id=2, return_step_logits=True, use_graph=False)
    S = emb_ref.shape[1]

    def dense(m):
        W = m.W if hasattr(m, "W") else m.weight
        return [_raw(W), _raw(m.bias)]

    def ln(m):
        return [_raw(m.weight), _raw(m.bias)]

    def qformer(bert, query, enc_ln, mlp):
        out = ln(enc_ln) + ln(bert.embeddings.LayerNorm) + [_raw(query[0])]
        for L in bert.encoder.layer:
            sa, ca = getattr(L.attention, "self"), getattr(L.crossattention, "self")
            out += dense(sa.query) + dense(sa._kv) + dense(L.attention.output.dense) + ln(L.attention.output.LayerNorm)
            out += dense(ca.query) + dense(ca._kv) + dense(L.crossattention.output.dense) + ln(L.crossattention.output.LayerNorm)
            out += dense(L.intermediate_query.dense) + dense(L.output_query.dense) + ln(L.output_query.LayerNorm)
        return out + dense(mlp[0]) + dense(mlp[2])

    cc, bc, qc, dc = meta["clip"], meta["beats"], meta["qf"], meta["dec"]
    Lc = max(meta["select"])
    sp = model.SPECIAL_TOKEN_2_IDS
    lc = __import__("crab_amd.peft_hyper", fromlist=["LoraConfig"]).LoraConfig()
    cfg = [dc["hidden_size"], dc["intermediate_size"], dc["num_hidden_layers"], dc["num_attention_heads"], dc["num_key_value_heads"],
           um.lm_head.weight.shape[0], lc.lora_nums, lc.r, n_new, 0,
           cc["hidden_size"], cc["intermediate_size"], Lc, cc["num_attention_heads"], cc["image_size"], cc["patch_size"],
           bc["embed_dim"], bc["encoder_embed_dim"], bc["encoder_ffn_embed_dim"], bc["encoder_layers"], bc["encoder_attention_heads"],
           bc["input_patch_size"], bc["conv_pos"], bc["conv_pos_groups"], bc["num_buckets"], bc["max_distance"], 128,
           qc["hidden"], qc["heads"], qc["inter"], 2, 32,
           p["t_v"], p["t_a"], p["l_a"], int(ids0.numel()), sp["<video>"], sp["<audio>"]]
    blob = [np.asarray(cfg, dtype=np.int32).tobytes(), mods['<video>'].float().contiguous().numpy().tobytes(),
            mods['<audio>'].float().contiguous().numpy().tobytes(), ids0.to(torch.int64).numpy().tobytes()]
    vt = inner.visual_encoder.vision_tower
    vm = vt.vision_model
    blob += [_raw(vt._patch_weight()), _raw(vm.embeddings.class_embedding), _raw(vm.embeddings.position_embedding.weight)] + ln(vm.pre_layrnorm)
    for L in vm.encoder.layers[:Lc]:
        blob += ln(L.layer_norm1) + dense(L.self_attn._qkv) + dense(L.self_attn.out_proj) + ln(L.layer_norm2) + dense(L.mlp.fc1) + dense(L.mlp.fc2)
    vl = inner.vl_projector
    blob += qformer(vl.visual_Qformer.bert, vl.visual_query_tokens, vl.visual_ln, vl.visual_proj)
    be = inner.audio_encoder.audio_encoder
    P_b = bc["input_patch_size"]
    blob += [_raw(be.patch_embedding.weight.reshape(bc["embed_dim"], P_b * P_b))] + ln(be.layer_norm) + dense(be.post_extract_proj)
    blob += [_raw(be._posconv_weight()), _raw(be.encoder.pos_conv[0].bias)] + ln(be.encoder.layer_norm)
    blob += [_raw(be.encoder.layers[0].self_attn.relative_attention_bias.weight)]
    for L in be.encoder.layers:
        a = L.self_attn
        blob += dense(a._qkv) + dense(a.out_proj) + dense(L.fc1) + dense(L.fc2) + ln(L.self_attn_layer_norm) + ln(L.final_layer_norm)
        blob += [_raw(a.grep_linear.weight), _raw(a.grep_linear.bias), _raw(a.grep_a.reshape(-1))]
    al = inner.al_projector
    blob += qformer(al.audio_Qformer.bert, al.audio_query_tokens, al.audio_ln, al.audio_proj)
    blob += [_raw(um.model.embed_tokens.weight)]
    for layer in um.model.layers:
        for g in layer.groups():
            blob.append(_raw(g.W))
            if g.bias is not None:
                blob.append(_raw(g.bias))
            blob += [_raw(g.RA), _raw(g.B2)]
        blob += [_raw(layer.input_layernorm.weight), _raw(layer.post_attention_layernorm.weight)]
    blob += [_raw(um.model.norm.weight), _raw(um.lm_head.weight), _raw(um.model.embed_tokens.weight)]
    bpath, exe = str(tmp_path / "clip.bin"), _build(tmp_path, "clip_demo")
    with open(bpath, "wb") as f:
        f.write(b"".join(blob))
    V = um.lm_head.weight.shape[0]
    ref_ids_np, ref_logits_np = ref_ids.cpu().numpy(), ref_logits.float().cpu().numpy()
    for use_graph in (0, 1):
        opath, epath = str(tmp_path / f"out{use_graph}.bin"), str(tmp_path / f"emb{use_graph}.bin")
        r = subprocess.run([exe, bpath, opath, epath, str(use_graph)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout, r.stderr)
        emb = torch.from_numpy(np.frombuffer(open(epath, "rb").read(), dtype=np.int16).copy()).view(BF).view(S, -1)
        assert torch.equal(emb, emb_ref[0].cpu()), float((emb.float() - emb_ref[0].cpu().float()).abs().max())
        raw = open(opath, "rb").read()
        ids = np.frombuffer(raw[:n_new * 8], dtype=np.int64).reshape(1, n_new)
        logits = np.frombuffer(raw[n_new * 8:], dtype=np.float32).reshape(n_new, 1, V).transpose(1, 0, 2)
        assert np.array_equal(ids, ref_ids_np), (use_graph, ids, ref_ids_np)
        assert np.array_equal(logits, ref_logits_np), (use_graph, np.abs(logits - ref_logits_np).max())
    assert np.array_equal(ref_ids_np, A["ids_bs1"].numpy()), "the C-run clip does not reproduce the ids the reference recorded"


def test_crab_gather_results_over_rccl_single_rank():
    """crab_dist_unique_id / crab_dist_init / crab_gather_results / crab_dist_destroy (csrc/dist.hip): the C-ABI form of the per-clip result
    gather, RCCL dlopen()ed on first use.  One GPU box: a world of one rank (the communicator, the stream-ordered ncclGather and
    the record layout are the real ones; the 2-rank form needs two devices and is exercised by the test below when they exist)."""
    import ctypes as C
    import torch
    from crab_amd import _lib
    lib = _lib.load()
    ctx = _lib.ctx(0)
    uid = (C.c_char * 128)()
    _lib.check(lib.crab_dist_unique_id(ctx, uid), 0)
    assert any(bytes(uid))
    comm = C.c_void_p()
    _lib.check(lib.crab_dist_init(ctx, uid, 1, 0, C.byref(comm)), 0)
    assert lib.crab_dist_world(comm) == 1 and lib.crab_dist_rank(comm) == 0
    n_new = 6
    rec = torch.stack([torch.cat([torch.tensor([cid]), torch.arange(n_new) + 100 * cid]) for cid in (4, 5, 6)]).to("cuda")      # {clip id, ids[n_new]} x 3 clips
    out = torch.zeros_like(rec)
    stream = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.crab_gather_results(ctx, stream, comm, rec.data_ptr(), rec.numel() * 8, out.data_ptr(), 0), 0)
    torch.cuda.synchronize()
    assert torch.equal(out, rec)
    assert lib.crab_gather_results(ctx, stream, comm, rec.data_ptr(), 0, out.data_ptr(), 0) < 0 and b"gather_results" in lib.crab_last_error(ctx)
    lib.crab_dist_destroy(comm)


def _gather_worker(rank, world, idfile, q):
    import ctypes as C
    import os
    import time
    import torch
    torch.cuda.set_device(rank)
    from crab_amd import _lib
    lib = _lib.load()
    ctx = _lib.ctx(rank)
    uid = (C.c_char * 128)()
    if rank == 0:
        _lib.check(lib.crab_dist_unique_id(ctx, uid), rank)
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(uid))
        os.replace(idfile + ".tmp", idfile)                       # the caller ships the id: here through a file
    else:
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.1)
        C.memmove(uid, open(idfile, "rb").read(), 128)
    comm = C.c_void_p()
    _lib.check(lib.crab_dist_init(ctx, uid, world, rank, C.byref(comm)), rank)
    rec = (torch.arange(5, dtype=torch.int64) + 1000 * rank).to(f"cuda:{rank}")
    out = torch.zeros(world * 5, dtype=torch.int64, device=f"cuda:{rank}") if rank == 0 else None
    _lib.check(lib.crab_gather_results(ctx, torch.cuda.current_stream().cuda_stream, comm, rec.data_ptr(), 40, out.data_ptr() if out is not None else None, 0), rank)
    torch.cuda.synchronize()
    if rank == 0:
        q.put(out.cpu().tolist())
    lib.crab_dist_destroy(comm)


def test_crab_gather_results_two_ranks_when_two_gpus_are_visible(tmp_path):
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU visible: the 2-rank RCCL gather needs two devices (the single-rank form runs above)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, str(tmp_path / "uid"), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got == list(range(5)) + [1000 + i for i in range(5)]


def test_attention_key_mask_against_dense_reference():
    """crab_attn_desc.key_mask / crab_attn_decode_keymask: one visibility bit per key (any 2-D attention_mask) in both prefill kernels
    (64-row and 128-row blocks, head_dim 64 and 128) and the decode kernel, against fp32 torch attention with the same mask; a query
    row without a visible key returns zeros."""
    import math
    from crab_amd import ops
    torch.manual_seed(5)
    for (B, H, Hk, S, d) in [(2, 4, 4, 200, 128), (3, 4, 2, 50, 64), (2, 2, 2, 257, 64)]:
        q = (torch.randn(B, S, H, d, device="cuda") * 0.5).to(torch.bfloat16)
        k = (torch.randn(B, Hk, S, d, device="cuda") * 0.5).to(torch.bfloat16)
        v = (torch.randn(B, Hk, S, d, device="cuda") * 0.5).to(torch.bfloat16)
        Sp = (S + 7) // 8 * 8
        vt = torch.zeros(B, Hk, d, Sp, device="cuda", dtype=torch.bfloat16); vt[..., :S] = v.transpose(2, 3)
        mask = torch.rand(B, S, device="cuda") > 0.3
        mask[0, :3] = False                                       # rows 0..2 of sequence 0 see nothing
        mask[:, S // 2] = False
        km = ops.pack_key_mask(mask)
        o = torch.empty(B, S, H * d, device="cuda", dtype=torch.bfloat16)
        scale = 1.0 / math.sqrt(d)
        ops.attn_fwd(q, k, vt, o, q_strides=(S * H * d, d, H * d), k_strides=(Hk * S * d, S * d, d), vt_strides=(Hk * d * Sp, d * Sp, Sp),
                     o_strides=(S * H * d, H * d), B=B, H=H, Hk=Hk, Sq=S, Skv=S, head_dim=d, scale=scale, causal=True, key_mask=km)
        kk = k.float().repeat_interleave(H // Hk, 1); vv = v.float().repeat_interleave(H // Hk, 1)
        sc = torch.einsum("bshd,bhtd->bhst", q.float(), kk) * scale
        allow = torch.tril(torch.ones(S, S, device="cuda", dtype=torch.bool))[None, None] & mask[:, None, None, :]
        sc = sc.masked_fill(~allow, float("-inf"))
        p = torch.softmax(sc, -1).nan_to_num(0.0)
        ref = torch.einsum("bhst,bhtd->bshd", p, vv).reshape(B, S, H * d)
        assert float((o.float() - ref).abs().max()) < 2e-2, (B, H, Hk, S, d)
        assert float(o[0, :3].float().abs().max()) == 0.0
        # decode: the last query row against the cache [B, Hk, Tmax, d] with ctx = S keys
        qd = q[:, -1].reshape(B, H * d).contiguous(); od = torch.empty_like(qd)
        ops.attn_decode(qd, k, v, od, B, H, Hk, d, S, S, scale, key_mask=km)
        assert float((od.float() - ref[:, -1]).abs().max()) < 2e-2
        pos = torch.full((1,), S - 1, device="cuda", dtype=torch.int32)       # context length as a device word: ctx = 1 + pos
        od2 = torch.empty_like(qd)
        ops.attn_decode(qd, k, v, od2, B, H, Hk, d, S, 1, scale, ctx_dev=pos, key_mask=km)
        assert torch.equal(od, od2)
    with pytest.raises(Exception, match="key_mask"):
        ops.attn_fwd(q, k, vt, o, q_strides=(S * H * d, d, H * d), k_strides=(Hk * S * d, S * d, d), vt_strides=(Hk * d * Sp, d * Sp, Sp),
                     o_strides=(S * H * d, H * d), B=B, H=H, Hk=Hk, Sq=S, Skv=S, head_dim=d, scale=scale, causal=True, key_mask=km[:, :2].contiguous())


def test_c_caller_of_the_segmentation_metrics_matches_the_python_mirror(tmp_path):
    """examples/metrics_demo.c: crab_mask_iou / crab_fmeasure / crab_miou_fscore called from C on buffers it uploads itself; every output array
    equals what crab_amd.avss_utils returns for the same masks, bit for bit (same library, same launches), and the CPU restatement's counts."""
    from crab_amd import avss_utils as AU
    from oracle import metrics_oracle as MO
    exe = str(tmp_path / "metrics_demo")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           os.path.join(ROOT, "examples", "metrics_demo.c"), "-o", exe, "-L" + os.path.join(ROOT, "crab_amd"), "-lcrab_hip", "-L/opt/rocm/lib",
                           "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "crab_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(9)
    N, H, W, T, BF, C, h, w = 3, 37, 53, 255, 2, 71, 32, 48
    pred = (rng.standard_normal((N, H, W)) * 3).astype(np.float32)
    gt = (rng.random((N, H, W)) > 0.5).astype(np.float32)
    gt[1] = 0
    th = AU.fmeasure_thresholds(T)
    cp = rng.standard_normal((BF, C, h, w)).astype(np.float32)
    ct = rng.integers(-1, C + 2, (BF, h, w)).astype(np.int64)
    with open(tmp_path / "in.bin", "wb") as f:
        for a in (pred, gt, th, cp, ct):
            f.write(a.tobytes())
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")] + [str(v) for v in (N, H * W, T, BF, C, h * w)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "fmeasure" in r.stdout and "T <= 1024" in r.stdout                   # the rejected call left its message in crab_last_error
    raw = open(tmp_path / "out.bin", "rb").read()
    off = 0

    def take(shape, dt):
        nonlocal off
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        a = np.frombuffer(raw[off:off + n], dt).reshape(shape)
        off += n
        return a
    counts, out2 = take((N, 6), np.int32), take((2,), np.float32)
    ge, ysum, fscore, score, best = take((N, 2, T), np.int32), take((N, 2), np.int32), take((N, T), np.float32), take((T,), np.float32), take((2,), np.float32)
    areas, iou_fc = take((BF, 3, C), np.int32), take((BF, C), np.float32)
    ious, fsc, cls, vid = take((C,), np.float32), take((C,), np.float32), take((C,), np.float32), take((BF,), np.float32)
    assert off == len(raw)
    P, G = torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda()
    iou, pc = AU.mask_iou(P, G, details=True)
    assert np.array_equal(counts, pc.numpy()) and out2[0] == iou.item() and out2[1] == AU.metric_s_for_null(P).item()
    assert np.array_equal(counts[:, :5], MO.mask_counts(pred, gt))
    val, d = AU.Eval_Fmeasure(P, G, details=True)
    assert np.array_equal(ge, d["ge"].numpy()) and np.array_equal(ysum, d["ysum"].numpy()) and np.array_equal(fscore, d["fscore"].numpy())
    assert np.array_equal(score, d["score"].numpy()) and best[0] == val and best[1] == 2
    mi, fs, cc, vd, dd = AU.calc_color_miou_fscore(torch.from_numpy(cp).cuda(), torch.from_numpy(ct).cuda(), details=True)
    assert np.array_equal(areas, dd["areas"].numpy()) and np.array_equal(areas, MO.class_areas(cp, ct)) and np.array_equal(iou_fc, dd["iou_fc"].numpy())
    assert np.array_equal(ious, mi.cpu().numpy()) and np.array_equal(fsc, fs.cpu().numpy()) and np.array_equal(cls, cc.cpu().numpy())
    assert np.array_equal(vid, torch.stack(vd).cpu().numpy(), equal_nan=True)


def test_library_loaded_before_torch_still_shares_its_hip_runtime():
    """r06: `build()` followed by `smoke()` in ONE process failed on the GPU box - build() loaded libcrab_hip.so before anything had imported torch, the
    library bound /opt/rocm's libamdhip64 and torch then brought its own: two HIP runtimes, crab_ctx_create saw no device while torch did.  _lib.load()
    now imports torch first; in a fresh interpreter the library-first order must work end to end."""
    import subprocess
    import sys
    code = ("from crab_amd import _lib\n"
            "import sys\n"
            "assert 'torch' not in sys.modules\n"
            "lib = _lib.load()\n"
            "assert 'torch' in sys.modules\n"
            "assert _lib.ctx(0)\n"
            "import torch\n"
            "from crab_amd import ops\n"
            "x = torch.randn(4, 64, device='cuda')\n"
            "y = ops.cast_bf16(x)\n"
            "assert torch.equal(y, x.to(torch.bfloat16))\n"
            "print('OK')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
