"""GPU parity tests of the individual HIP kernels, called through the C-ABI (crab_amd.ops -> ctypes).

Reference for each op: fp32 CPU arithmetic on the SAME bf16-rounded inputs (oracle/crab_oracle.py where a
restatement exists, otherwise the plain torch fp32 op).  Tolerances are set to at most twice the error measured on the
GPU (every comparison is recorded in the parity report, tests/util.py:record_parity)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16
# Tolerances = at most 2x the worst error measured on MI355X (profiles/r02_parity_report.json), relative to max |reference|:
TOL_F32 = 3e-6       # fp32 outputs: accumulation order only (worst 1.15e-6 at K = 6144)
TOL_BF16 = 6e-3      # bf16 outputs: one storage rounding on top (worst 3.0e-3); attention kernels 2.0-3.0e-3


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def _cmp(got, ref, rel=2e-2, what=""):
    from tests.util import rel_err
    r = rel_err(got, ref, what, rel)
    assert r <= rel, f"{what}: relative max err {r:.4g} > {rel:.4g}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (16, 16, 32), (130, 200, 72), (702, 512, 1024), (257, 136, 592),
                                   (64, 4104, 256), (300, 48, 6144), (1, 320, 128), (2056, 1024, 1024)])
def test_gemm_shapes_fp32_out(M, N, K):
    from crab_amd import ops
    x, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=1 / math.sqrt(K))
    y = ops.gemm(x.cuda(), w.cuda(), out_fp32=True)
    _cmp(y, x.float() @ w.float().t(), TOL_F32, f"gemm {M}x{N}x{K}")


@pytest.mark.parametrize("act", ["none", "gelu", "quick_gelu", "relu", "silu"])
def test_gemm_epilogue(act):
    from crab_amd import ops
    M, N, K = 200, 264, 328
    x, w, b, r = _rand(M, K, seed=3), _rand(N, K, seed=4, scale=1 / math.sqrt(K)), _rand(N, seed=5), _rand(M, N, seed=6)
    y = ops.gemm(x.cuda(), w.cuda(), bias=b.cuda(), act=act, residual=r.cuda(), res_scale=2.2133)
    z = x.float() @ w.float().t() + b.float()
    z = {"none": z, "gelu": F.gelu(z), "quick_gelu": z * torch.sigmoid(1.702 * z), "relu": F.relu(z), "silu": F.silu(z)}[act]
    _cmp(y, z + 2.2133 * r.float(), TOL_BF16, act)
    assert y.dtype == BF


def test_gemm_second_k_segment_and_odd_ldc():
    from crab_amd import ops
    M, N, K, K2 = 70, 321, 128, 32           # N odd -> scalar store path
    x, w = _rand(M, K, seed=7), _rand(N, K, seed=8, scale=0.1)
    x2, w2 = _rand(M, K2, seed=9), _rand(N, K2, seed=10, scale=0.1)
    y = ops.gemm(x.cuda(), w.cuda(), x2=x2.cuda(), w2=w2.cuda(), out_fp32=True)
    _cmp(y, x.float() @ w.float().t() + x2.float() @ w2.float().t(), TOL_F32, "2-seg")


def test_gemm_batched_sliding_window():
    """BEATs pos-conv form: overlapping A rows (lda < K), two-level batch strides."""
    import ctypes as C
    from crab_amd import ops, _lib
    G, B, n, cg, Kc = 4, 3, 20, 8, 16
    np_ = n + Kc - 1
    xp = _rand(G, B, np_, cg, seed=11)
    w = _rand(G, cg, Kc * cg, seed=12, scale=0.1)       # [G][co][(k,ci)]
    bias = _rand(G * cg, seed=13)
    res = _rand(B, n, G * cg, seed=14)
    out = torch.empty(B, n, G * cg, dtype=BF, device="cuda")
    xp_d, w_d, b_d, r_d = xp.cuda(), w.cuda(), bias.cuda(), res.cuda()
    g = _lib.GemmDesc()
    g.A, g.B, g.C, g.bias, g.R = xp_d.data_ptr(), w_d.data_ptr(), out.data_ptr(), b_d.data_ptr(), r_d.data_ptr()
    g.lda, g.ldb, g.ldc, g.ldr = cg, Kc * cg, G * cg, G * cg
    g.M, g.N, g.K = n, cg, Kc * cg
    g.act, g.c_fp32, g.res_scale = 1, 0, 1.0
    g.batch, g.nb0 = G * B, B                     # z0 = b, z1 = g
    g.sA0, g.sA1 = np_ * cg, B * np_ * cg
    g.sB0, g.sB1 = 0, cg * Kc * cg
    g.sC0, g.sC1 = n * G * cg, cg
    g.sR0, g.sR1 = n * G * cg, cg
    g.sBias0, g.sBias1 = 0, cg
    ops.gemm_desc(g)
    ref = torch.empty(B, n, G * cg)
    for gi in range(G):
        for b in range(B):
            A = torch.stack([xp[gi, b, t:t + Kc].reshape(-1) for t in range(n)]).float()
            z = F.gelu(A @ w[gi].float().t() + bias[gi * cg:(gi + 1) * cg].float())
            ref[b, :, gi * cg:(gi + 1) * cg] = z + res[b, :, gi * cg:(gi + 1) * cg].float()
    _cmp(out, ref, TOL_BF16, "sliding-window batched gemm")


def test_rmsnorm_layernorm():
    from crab_amd import ops
    from oracle import crab_oracle as O
    for D in (128, 1024, 4096, 3584):
        x, w, b = _rand(37, D, seed=D), (1 + 0.1 * torch.randn(D)).to(BF), (0.1 * torch.randn(D)).to(BF)
        y = ops.rmsnorm(x.cuda(), w.cuda(), 1e-5)
        _cmp(y, O.rmsnorm(x.float(), w.float(), 1e-5), 1e-2, f"rmsnorm {D}")
        y = ops.layernorm(x.cuda(), w.cuda(), b.cuda(), 1e-5)
        _cmp(y, F.layer_norm(x.float(), (D,), w.float(), b.float(), 1e-5), 1e-2, f"layernorm {D}")
        # LayerNorm parameters in fp32 (crab_layernorm_p: what the encoder modules hold since r05), bf16 and fp32 input rows: the only
        # rounding left is the bf16 output
        g = torch.Generator().manual_seed(D)
        w32, b32 = 1 + 0.1 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
        xf = torch.randn(37, D, generator=g)
        ref = F.layer_norm(x.float(), (D,), w32, b32, 1e-5)
        y = ops.layernorm(x.cuda(), w32.cuda(), b32.cuda(), 1e-5)
        _cmp(y, ref, 4.5e-3, f"layernorm {D}, fp32 parameters")
        assert torch.equal(y.cpu(), ref.to(BF)) or (y.cpu().float() - ref.to(BF).float()).abs().max() <= 2 * ref.abs().max() * 2 ** -8
        y = ops.layernorm(xf.cuda(), w32.cuda(), b32.cuda(), 1e-5)
        _cmp(y, F.layer_norm(xf, (D,), w32, b32, 1e-5), 4.5e-3, f"layernorm {D}, fp32 rows and fp32 parameters")
        y = ops.layernorm(xf.cuda(), w32.cuda(), None, 1e-5)
        _cmp(y, F.layer_norm(xf, (D,), w32, None, 1e-5), 4.5e-3, f"layernorm {D}, fp32 rows, fp32 weight, no bias")
        with pytest.raises(Exception):
            ops.layernorm(x.cuda(), w32.cuda(), b.cuda(), 1e-5)          # weight and bias must share their storage


def test_embedding_swiglu_argmax_cast():
    from crab_amd import ops
    tab = _rand(50, 128, seed=1)
    ids = torch.tensor([3, 49, 0, 7, 7])
    assert torch.equal(ops.embedding(ids.cuda(), tab.cuda()).cpu(), tab[ids])
    gu = _rand(33, 2 * 264, seed=2)
    _cmp(ops.swiglu(gu.cuda()), F.silu(gu[:, :264].float()) * gu[:, 264:].float(), TOL_BF16, "swiglu")
    lg = torch.randn(5, 32017)
    lg[2, 100] = lg[2, 5000] = 50.0          # tie -> first index
    assert torch.equal(ops.argmax(lg.cuda()).cpu(), lg.argmax(-1))
    sup = int(lg[0].argmax())
    l2 = lg.clone(); l2[:, sup] = -float("inf")
    assert torch.equal(ops.argmax(lg.cuda(), suppress=sup).cpu(), l2.argmax(-1))
    x = torch.randn(1000)
    assert torch.equal(ops.cast_bf16(x.cuda()).cpu(), x.to(BF))


def test_hyperlora_mix_matches_reference_formula():
    from crab_amd import ops
    M, nproj, nl, r = 19, 3, 3, 8
    t = torch.randn(M, 40)
    u = ops.hyperlora_mix(t.cuda(), nproj, nl, r, 96, 2.0).cpu().float()
    ref = torch.zeros(M, 96)
    for p in range(nproj):
        seg = t[:, p * 11:(p + 1) * 11]
        pr = torch.softmax(seg[:, :3], -1)
        for i in range(3):
            ref[:, p * 24 + i * 8:p * 24 + (i + 1) * 8] = 2.0 * pr[:, i:i + 1] * seg[:, 3:]
    _cmp(u, ref, TOL_BF16, "mix")
    assert (u[:, 72:] == 0).all()


def test_hyperlora_linear_fused_matches_golden():
    """Full fused path (skinny [R;A] GEMM -> mix -> two-segment GEMM) against the reference-generated fixture."""
    from crab_amd import ops
    from tests.util import load_fixture, weights_from_table
    meta, A = load_fixture("hyperlora_linear")
    W = {k: v.to(BF) for k, v in weights_from_table(meta).items()}
    x = A["x"].reshape(-1, 128).to(BF)
    ra = torch.cat([W["lin.lora_route.weight"], W["lin.lora_A.weight"]], 0)
    ra = torch.cat([ra, torch.zeros(16 - ra.shape[0], 128, dtype=BF)], 0)
    bcat = torch.cat([W[f"lin.lora_B{i}.weight"] for i in range(3)], 1)
    bcat = torch.cat([bcat, torch.zeros(256, 8, dtype=BF)], 1)
    t = ops.gemm(x.cuda(), ra.cuda(), out_fp32=True)
    u = ops.hyperlora_mix(t, 1, 3, 8, 32, 2.0)
    y = ops.gemm(x.cuda(), W["lin.weight"].cuda(), bias=W["lin.bias"].cuda(), x2=u, w2=bcat.cuda(), out_fp32=True)
    _cmp(y, A["y"].reshape(-1, 256), TOL_BF16, "hyper-LoRA linear vs reference")


def _attn_ref(q, k, v, scale, causal=False, bias=None):
    # q [B,H,Sq,d], k/v [B,Hk,Skv,d]
    B, H, Sq, d = q.shape
    Hk, Skv = k.shape[1], k.shape[2]
    g = H // Hk
    k = k[:, :, None].expand(B, Hk, g, Skv, d).reshape(B, H, Skv, d)
    v = v[:, :, None].expand(B, Hk, g, Skv, d).reshape(B, H, Skv, d)
    a = q.float() @ k.float().transpose(2, 3) * scale
    if bias is not None:
        a = a + bias
    if causal:
        i = torch.arange(Sq)[:, None] + (Skv - Sq)
        j = torch.arange(Skv)[None]
        a = a.masked_fill(j > i, float("-inf"))
    return torch.softmax(a, -1) @ v.float()


@pytest.mark.parametrize("hd,B,H,Hk,Sq,Skv,causal", [
    (128, 2, 4, 4, 150, 150, True), (128, 1, 4, 2, 702, 702, True), (64, 3, 2, 2, 257, 257, False),
    (64, 2, 12, 12, 32, 256, False), (64, 2, 2, 2, 32, 48, False), (128, 1, 2, 2, 1, 70, False), (64, 1, 4, 1, 65, 129, True),
    # the 128-rows-per-block kernel (Sq > 64): exact block / wave boundaries, ragged last key tile, causal offset, grouped kv heads
    (128, 1, 2, 2, 128, 128, True), (128, 2, 2, 1, 129, 129, True), (128, 1, 2, 2, 300, 300, False), (64, 2, 4, 2, 97, 200, True),
    (64, 1, 2, 2, 448, 449, False), (128, 1, 1, 1, 66, 66, True)])
def test_attn_fwd(hd, B, H, Hk, Sq, Skv, causal):
    from crab_amd import ops
    q, k, v = _rand(B, H, Sq, hd, seed=1), _rand(B, Hk, Skv, hd, seed=2), _rand(B, Hk, Skv, hd, seed=3)
    Sp = (Skv + 7) // 8 * 8
    vt = torch.zeros(B, Hk, hd, Sp, dtype=BF)
    vt[..., :Skv] = v.transpose(2, 3)
    # q laid out token-major [B,Sq,H*hd] like a projection output; k head-major like the KV cache
    qt = q.transpose(1, 2).reshape(B, Sq, H * hd).contiguous().cuda()
    kd, vtd = k.cuda(), vt.cuda()
    o = torch.zeros(B, Sq, H * hd, dtype=BF, device="cuda")
    scale = hd ** -0.5
    ops.attn_fwd(qt, kd, vtd, o, q_strides=(Sq * H * hd, hd, H * hd), k_strides=(Hk * Skv * hd, Skv * hd, hd),
                 vt_strides=(Hk * hd * Sp, hd * Sp, Sp), o_strides=(Sq * H * hd, H * hd), B=B, H=H, Hk=Hk, Sq=Sq, Skv=Skv,
                 head_dim=hd, scale=scale, causal=causal)
    ref = _attn_ref(q, k, v, scale, causal).transpose(1, 2).reshape(B, Sq, H * hd)
    _cmp(o, ref, TOL_BF16, "attn_fwd")


def test_attn_fwd_gated_bias():
    from crab_amd import ops
    B, H, n, hd = 3, 2, 48, 64
    q, k, v = _rand(B, H, n, hd, seed=1), _rand(B, H, n, hd, seed=2), _rand(B, H, n, hd, seed=3)
    bias, gate = torch.randn(H, n, n), 1 + torch.rand(B, H, n)
    vt = v.transpose(2, 3).contiguous()
    qt = q.transpose(1, 2).reshape(B, n, H * hd).contiguous().cuda()
    o = torch.zeros(B, n, H * hd, dtype=BF, device="cuda")
    ops.attn_fwd(qt, k.cuda(), vt.cuda(), o, q_strides=(n * H * hd, hd, H * hd), k_strides=(H * n * hd, n * hd, hd),
                 vt_strides=(H * hd * n, hd * n, n), o_strides=(n * H * hd, H * hd), B=B, H=H, Hk=H, Sq=n, Skv=n, head_dim=hd,
                 scale=hd ** -0.5, bias=bias.cuda(), gate=gate.cuda())
    ref = _attn_ref(q, k, v, hd ** -0.5, False, gate[..., None] * bias[None]).transpose(1, 2).reshape(B, n, H * hd)
    _cmp(o, ref, TOL_BF16, "gated-bias attention")


@pytest.mark.parametrize("hd,H,Hk", [(128, 4, 4), (64, 4, 2)])
def test_rope_split_and_decode_attention(hd, H, Hk):
    from crab_amd import ops
    from oracle import crab_oracle as O
    B, S, Tmax = 2, 37, 64
    theta = 10000.0
    qkv = _rand(B * S, (H + 2 * Hk) * hd, seed=5)
    tab = ops.rope_table(Tmax, hd, theta, "cuda")
    kc = torch.zeros(B, Hk, Tmax, hd, dtype=BF, device="cuda")
    vc = torch.zeros_like(kc)
    Sp = 40
    vt = torch.zeros(B, Hk, hd, Sp, dtype=BF, device="cuda")
    qd = qkv.cuda()
    ops.qkv_rope_split(qd, tab, kc, vc, vt, B, S, H, Hk, hd, Tmax, pos0=0)
    q = qkv[:, :H * hd].float().view(B, S, H, hd).transpose(1, 2)
    k = qkv[:, H * hd:(H + Hk) * hd].float().view(B, S, Hk, hd).transpose(1, 2)
    v = qkv[:, (H + Hk) * hd:].view(B, S, Hk, hd).transpose(1, 2)
    cos, sin = O.rope_cos_sin(torch.arange(S)[None].expand(B, S), hd, theta)
    qr, kr = O.apply_rope(q, k, cos, sin)
    _cmp(qd[:, :H * hd].view(B, S, H, hd).transpose(1, 2), qr, TOL_BF16, "rope q")
    _cmp(kc[:, :, :S], kr, TOL_BF16, "rope k -> cache")
    assert torch.equal(vc[:, :, :S].cpu(), v)
    assert torch.equal(vt[..., :S].cpu(), v.transpose(2, 3))
    # decode step at position S (device-resident position word)
    pos = torch.tensor([S], dtype=torch.int32, device="cuda")
    q1 = _rand(B, (H + 2 * Hk) * hd, seed=6)
    q1d = q1.cuda()
    ops.qkv_rope_split(q1d, tab, kc, vc, None, B, 1, H, Hk, hd, Tmax, pos0=0, pos_dev=pos)
    o = torch.zeros(B, H * hd, dtype=BF, device="cuda")
    ops.attn_decode(q1d, kc, vc, o, B, H, Hk, hd, Tmax, 1, hd ** -0.5, ctx_dev=pos)
    qn = q1d[:, :H * hd].cpu().view(B, 1, H, hd).transpose(1, 2)
    ref = _attn_ref(qn, kc[:, :, :S + 1].cpu(), vc[:, :, :S + 1].cpu(), hd ** -0.5).transpose(1, 2).reshape(B, H * hd)
    _cmp(o, ref, TOL_BF16, "decode attention")


@pytest.mark.parametrize("hd,B,H,Hk,S", [(128, 3, 4, 4, 200), (64, 3, 4, 2, 150), (128, 2, 8, 2, 77)])
def test_attention_left_pad_mask_and_explicit_rotary_positions(hd, B, H, Hk, S):
    """forward()'s attention_mask / position_ids (models/unified_llama.py:149-160) at the kernel level: kv_start (first visible key
    per sequence, several 64-key tiles deep) in the causal flash forward and in the decode attention, and per-token rotary positions
    in the RoPE / KV-split pass, against fp32 arithmetic.  Valid rows only; pad rows must be finite (zeros)."""
    from crab_amd import ops
    from oracle import crab_oracle as O
    Tmax, theta = 256, 10000.0
    starts = [0, 70, 131][:B] if B == 3 else [5, 0]
    st = torch.tensor(starts, dtype=torch.int32)
    pos_ids = (torch.arange(S)[None] - st[:, None].long()).clamp(min=0).to(torch.int32)            # cumsum(mask) - 1, pads -> 0
    qkv = _rand(B * S, (H + 2 * Hk) * hd, seed=15)
    tab = ops.rope_table(Tmax, hd, theta, "cuda")
    kc = torch.zeros(B, Hk, Tmax, hd, dtype=BF, device="cuda")
    vc = torch.zeros_like(kc)
    Sp = (S + 7) // 8 * 8
    vt = torch.zeros(B, Hk, hd, Sp, dtype=BF, device="cuda")
    qd = qkv.cuda()
    ops.qkv_rope_split(qd, tab, kc, vc, vt, B, S, H, Hk, hd, Tmax, pos0=0, pos_ids=pos_ids.cuda())
    q = qkv[:, :H * hd].float().view(B, S, H, hd).transpose(1, 2)
    k = qkv[:, H * hd:(H + Hk) * hd].float().view(B, S, Hk, hd).transpose(1, 2)
    v = qkv[:, (H + Hk) * hd:].view(B, S, Hk, hd).transpose(1, 2)
    cos, sin = O.rope_cos_sin(pos_ids.long(), hd, theta)
    qr, kr = O.apply_rope(q, k, cos, sin)
    _cmp(qd[:, :H * hd].view(B, S, H, hd).transpose(1, 2), qr, TOL_BF16, "rope q at explicit positions")
    _cmp(kc[:, :, :S], kr, TOL_BF16, "rope k at explicit positions -> cache slot s")
    o = torch.full((B, S, H * hd), float("nan"), dtype=BF, device="cuda")
    ldq = qd.stride(0)
    ops.attn_fwd(qd, kc, vt, o, q_strides=(S * ldq, hd, ldq), k_strides=(Hk * Tmax * hd, Tmax * hd, hd), vt_strides=(Hk * hd * Sp, hd * Sp, Sp),
                 o_strides=(S * H * hd, H * hd), B=B, H=H, Hk=Hk, Sq=S, Skv=S, head_dim=hd, scale=hd ** -0.5, causal=True, kv_start=st.cuda())
    assert torch.isfinite(o.float()).all()
    qb = qd[:, :H * hd].cpu().view(B, S, H, hd).transpose(1, 2)
    for b in range(B):
        s0 = starts[b]
        ref = _attn_ref(qb[b:b + 1, :, s0:], kc[b:b + 1, :, s0:S].cpu(), vc[b:b + 1, :, s0:S].cpu(), hd ** -0.5, causal=True)
        _cmp(o[b, s0:].view(S - s0, H, hd).transpose(0, 1)[None], ref, TOL_BF16, f"left-pad masked causal attention, row {b} (kv_start {s0})")
        assert float(o[b, :s0].float().abs().max()) == 0.0 if s0 else True          # pad query rows: zeros
    # decode step at cache slot S, rotary position S - start, keys [start, S]
    pos = torch.tensor([S], dtype=torch.int32, device="cuda")
    q1 = _rand(B, (H + 2 * Hk) * hd, seed=16).cuda()
    p1 = (S - st.long())[:, None].to(torch.int32)
    ops.qkv_rope_split(q1, tab, kc, vc, None, B, 1, H, Hk, hd, Tmax, pos0=0, pos_dev=pos, pos_ids=p1.cuda())
    o1 = torch.zeros(B, H * hd, dtype=BF, device="cuda")
    ops.attn_decode(q1, kc, vc, o1, B, H, Hk, hd, Tmax, 1, hd ** -0.5, ctx_dev=pos, kv_start=st.cuda())
    for b in range(B):
        s0 = starts[b]
        qn = q1[b:b + 1, :H * hd].cpu().view(1, 1, H, hd).transpose(1, 2)
        ref = _attn_ref(qn, kc[b:b + 1, :, s0:S + 1].cpu(), vc[b:b + 1, :, s0:S + 1].cpu(), hd ** -0.5).transpose(1, 2).reshape(1, H * hd)
        _cmp(o1[b:b + 1], ref, TOL_BF16, f"left-pad masked decode attention, row {b}")


@pytest.mark.parametrize("ctx", [702, 830, 958])
def test_decode_attention_mha_benchmark_regime(ctx):
    """attn_decode_kernel<128> exactly as bench.py launches it (the dominant kernel of the benchmark): B = 256 clips, H = Hk = 32,
    head_dim 128, Tmax = 960, live context read from the device position word; EVERY (clip, head) row against fp32 arithmetic on the
    same bf16 K / V / q.  Cache rows at and beyond the live context are poisoned: reading one would move the result by orders of
    magnitude."""
    from crab_amd import ops
    B, H, hd, Tmax = 256, 32, 128, 960
    g = torch.Generator(device="cuda").manual_seed(1000 + ctx)
    kc = torch.empty(B, H, Tmax, hd, device="cuda", dtype=BF)
    vc = torch.empty_like(kc)
    for b0 in range(0, B, 32):                              # generated in slices: the fp32 temporaries stay small
        kc[b0:b0 + 32] = (torch.randn(32, H, Tmax, hd, device="cuda", generator=g) * 0.7).to(BF)
        vc[b0:b0 + 32] = (torch.randn(32, H, Tmax, hd, device="cuda", generator=g) * 0.7).to(BF)
    kc[:, :, ctx:] = 3.0e4
    vc[:, :, ctx:] = -3.0e4
    q = (torch.randn(B, 3 * H * hd, device="cuda", generator=g) * 1.5).to(BF)      # packed q|k|v row, as the decode step passes it
    pos = torch.tensor([ctx - 1], dtype=torch.int32, device="cuda")               # position of the newest token: ctx live rows
    o = torch.zeros(B, H * hd, dtype=BF, device="cuda")
    ops.attn_decode(q, kc, vc, o, B, H, H, hd, Tmax, 1, hd ** -0.5, ctx_dev=pos)
    ref = torch.empty(B, H * hd, dtype=torch.float32, device="cuda")
    for b0 in range(0, B, 32):
        qf = q[b0:b0 + 32, :H * hd].float().view(32, H, 1, hd)
        a = torch.matmul(qf, kc[b0:b0 + 32, :, :ctx].float().transpose(2, 3)) * hd ** -0.5
        ref[b0:b0 + 32] = torch.matmul(torch.softmax(a, -1), vc[b0:b0 + 32, :, :ctx].float()).reshape(32, H * hd)
    _cmp(o, ref.cpu(), TOL_BF16, f"decode attention MHA, B=256 H=32 ctx={ctx} (bench regime)")
    # determinism of the launch the benchmark replays from its HIP graph
    o2 = torch.zeros_like(o)
    ops.attn_decode(q, kc, vc, o2, B, H, H, hd, Tmax, 1, hd ** -0.5, ctx_dev=pos)
    assert torch.equal(o, o2)


@pytest.mark.parametrize("G,ctx", [(7, 830), (7, 1100), (4, 333), (2, 64), (8, 1025)])
def test_decode_attention_grouped_query(G, ctx):
    """GQA decode kernel (one block per (b, kv head), K/V rows read once for the G query heads) vs fp32 arithmetic and vs
    the per-(b,h) kernel it replaces (taken when B*Hk < 256)."""
    from crab_amd import ops
    hd, Hk, B, Tmax = 128, 4, 64, 1152
    H = G * Hk
    g = torch.Generator().manual_seed(100 + G)
    kc = (torch.randn(B, Hk, Tmax, hd, generator=g) * 0.7).to(BF).cuda()
    vc = (torch.randn(B, Hk, Tmax, hd, generator=g) * 0.7).to(BF).cuda()
    q = (torch.randn(B, H * hd, generator=g) * 1.5).to(BF).cuda()
    pos = torch.tensor([ctx - 1], dtype=torch.int32, device="cuda")
    o = torch.zeros(B, H * hd, dtype=BF, device="cuda")
    ops.attn_decode(q, kc, vc, o, B, H, Hk, hd, Tmax, 1, hd ** -0.5, ctx_dev=pos)          # B*Hk = 256 -> grouped kernel
    qn = q.cpu().view(B, 1, H, hd).transpose(1, 2)
    ref = _attn_ref(qn, kc[:, :, :ctx].cpu(), vc[:, :, :ctx].cpu(), hd ** -0.5).transpose(1, 2).reshape(B, H * hd)
    _cmp(o, ref, TOL_BF16, "grouped decode attention")
    o2 = torch.zeros(3, H * hd, dtype=BF, device="cuda")
    ops.attn_decode(q[:3].contiguous(), kc[:3].contiguous(), vc[:3].contiguous(), o2, 3, H, Hk, hd, Tmax, ctx, hd ** -0.5)   # per-head kernel
    _cmp(o[:3], o2.float().cpu(), TOL_BF16, "grouped vs per-head decode kernel")


@pytest.mark.parametrize("V,T,k,p", [(300, 0.6, 50, 0.9), (32017, 0.6, 50, 0.9), (1000, 1.3, 7, 0.5), (64, 0.8, 0, 0.95), (40, 1.0, 5, 1.0)])
def test_sample_select_draws_from_the_hf_distribution(V, T, k, p):
    """crab_sample_select (temperature -> top-k -> top-p -> draw, device-resident) against the distribution HF's sample mode draws from
    (oracle.sampling_probs, pinned to transformers' warpers on CPU): 16384 independent rows with the same logits - no draw outside the
    kept set, total-variation distance to the reference probabilities < 3 %, every kept token with p >= 2 % is drawn; deterministic per
    (seed, step); top_k = 1 is the argmax; EOS suppression below min_new_tokens."""
    from crab_amd import ops
    from oracle import crab_oracle as O
    g = torch.Generator().manual_seed(V)
    lg = torch.randn(1, V, generator=g) * 1.2
    probs = O.sampling_probs(lg, T, k, p)[0]
    assert int((probs > 0).sum()) >= 2, "test input too peaked: the kept set must hold several tokens"
    N = 16384
    logits = lg.expand(N, V).contiguous().cuda()

    def draw(seed, step, eos=-1, min_new=0, kk=k, pp=p, tt=T):
        cur = torch.zeros(N, dtype=torch.int64, device="cuda")
        out = torch.full((N, 4), -1, dtype=torch.int64, device="cuda")
        fin = torch.zeros(N, dtype=torch.int32, device="cuda")
        sd = torch.tensor([step], dtype=torch.int32, device="cuda")
        ops.sample_select(logits, cur, out, sd, fin, eos, 2, min_new, tt, kk, pp, seed)
        assert torch.equal(out[:, step], cur)
        return cur.cpu()
    a = draw(1234, 0)
    assert torch.equal(a, draw(1234, 0)), "not deterministic for a given (seed, step)"
    assert not torch.equal(a, draw(1234, 1)) and not torch.equal(a, draw(99, 0))
    hist = torch.bincount(a, minlength=V).float() / N
    assert float(hist[probs == 0].sum()) == 0.0, "a token outside HF's kept set was drawn"
    tv = 0.5 * float((hist - probs).abs().sum())
    from tests.util import record_parity
    record_parity(f"sample_select V={V} T={T} top_k={k} top_p={p}: total-variation distance of 16384 draws to the HF distribution", tv, 1.0, 3e-2,
                  kept_tokens=int((probs > 0).sum()))
    assert tv < 3e-2, tv
    assert bool((hist[probs >= 0.02] > 0).all())
    assert torch.equal(draw(7, 2, kk=1), torch.full((N,), int(lg.argmax()), dtype=torch.int64))
    top = int(lg.argmax())
    sup = draw(7, 0, eos=top, min_new=1)                      # the most likely token is EOS and still suppressed at step 0
    assert not bool((sup == top).any())


def test_im2col_and_clip_embed():
    from crab_amd import ops
    x = torch.randn(2, 3, 28, 42)
    p = ops.im2col_patch(x.cuda(), 14, 592).cpu().float()
    ref = F.unfold(x, 14, stride=14).transpose(1, 2).reshape(-1, 588)
    _cmp(p[:, :588], ref.to(BF), 1e-6, "im2col")
    assert (p[:, 588:] == 0).all()
    # BEATs shape: [B,1,L,128] with L=98 -> 6 time patches (rows 96,97 dropped)
    a = torch.randn(2, 1, 98, 128)
    pa = ops.im2col_patch(a.cuda(), 16, 256).cpu().float()
    assert pa.shape == (2 * 6 * 8, 256)
    _cmp(pa, F.unfold(a, 16, stride=16).transpose(1, 2).reshape(-1, 256).to(BF), 1e-6, "im2col beats")
    N, P, D = 2, 6, 128
    patch, cls, pos, w, b = _rand(N * P, D, seed=1), _rand(D, seed=2), _rand(P + 1, D, seed=3), _rand(D, seed=4), _rand(D, seed=5)
    y = ops.clip_embed_ln(patch.cuda(), cls.cuda(), pos.cuda(), w.cuda(), b.cuda(), N, P, D, 1e-5)
    xx = torch.cat([cls.float().expand(N, 1, D), patch.float().view(N, P, D)], 1) + pos.float()
    _cmp(y, F.layer_norm(xx, (D,), w.float(), b.float(), 1e-5).reshape(-1, D), TOL_BF16, "clip embed ln")
    # pre_layrnorm's parameters in fp32 (crab_clip_embed_ln_p, what the modules hold since r05), vector and scalar kernel (odd width)
    for D2 in (128, 100):
        g = torch.Generator().manual_seed(D2)
        patch, cls, pos = _rand(N * P, D2, seed=1), _rand(D2, seed=2), _rand(P + 1, D2, seed=3)
        w32, b32 = 1 + 0.1 * torch.randn(D2, generator=g), 0.1 * torch.randn(D2, generator=g)
        y = ops.clip_embed_ln(patch.cuda(), cls.cuda(), pos.cuda(), w32.cuda(), b32.cuda(), N, P, D2, 1e-5)
        xx = torch.cat([cls.float().expand(N, 1, D2), patch.float().view(N, P, D2)], 1) + pos.float()
        _cmp(y, F.layer_norm(xx, (D2,), w32, b32, 1e-5).reshape(-1, D2), TOL_BF16, f"clip embed ln, fp32 parameters, D={D2}")


def test_beats_helpers_match_golden_buckets():
    from crab_amd import ops
    from oracle import crab_oracle as O
    from tests.util import load_fixture
    meta, A = load_fixture("beats_buckets")
    H = 12
    table = torch.arange(320, dtype=torch.float32)[:, None].expand(320, H).contiguous().to(BF)   # exact in bf16 up to 256..
    table = (torch.arange(320) % 251).float()[:, None].expand(320, H).contiguous().to(BF)
    for n in (48, 96):
        bias = ops.beats_relpos_bias(table.cuda(), n, H, 320, 800).cpu()
        ref = (A[f"b{n}"].long() % 251).float()
        assert torch.equal(bias[0], ref) and torch.equal(bias[H - 1], ref)
    B, n, d = 2, 48, 64
    q, gw, gb, ga = _rand(B * n, H * d, seed=1), _rand(8, d, seed=2, scale=0.2), _rand(8, seed=3), (1 + 0.1 * torch.randn(H)).to(BF)
    gate = ops.beats_gru_gate(q.cuda(), gw.cuda(), gb.cuda(), ga.cuda(), B, n, H, d).cpu()
    qh = q.float().view(B, n, H, d).transpose(1, 2)
    gl = torch.sigmoid(F.linear(qh, gw.float(), gb.float()).view(B, H, n, 2, 4).sum(-1))
    ref = gl[..., 0] * (gl[..., 1] * ga.float().view(1, H, 1) - 1.0) + 2.0
    _cmp(gate, ref, 1e-6, "gru gate")
    x = _rand(B, n, 128, seed=9)
    xp = ops.beats_posconv_pad(x.cuda(), B, n, 128, 16, 128).cpu()
    ref = torch.zeros(16, B, n + 127, 8, dtype=BF)
    ref[:, :, 64:64 + n] = x.view(B, n, 16, 8).permute(2, 0, 1, 3)
    assert torch.equal(xp, ref)


@pytest.mark.parametrize("M,K,nproj", [(8, 4096, 3), (64, 11008, 1), (70, 128, 2), (2808, 4096, 3), (1, 256, 1), (256, 4096, 3), (256, 11008, 1),
                                       (250, 4096, 2), (257, 4096, 3), (4500, 4096, 3), (5000, 1056, 1)])
def test_hyperlora_route_matches_gemm_plus_mix(M, K, nproj):
    """Router (fused single launch for M <= 256, split-K pair above) == (fp32 product -> softmax mix), run-to-run deterministic."""
    from crab_amd import ops
    tcols = (nproj * 11 + 15) // 16 * 16
    ucols = (nproj * 24 + 31) // 32 * 32
    x = _rand(M, K, seed=1)
    ra = _rand(tcols, K, seed=2, scale=K ** -0.5)
    ra[nproj * 11:] = 0
    xd, rad = x.cuda(), ra.cuda()
    u1 = ops.hyperlora_route(xd, rad, nproj, 3, 8, ucols, 2.0)
    u2 = ops.hyperlora_route(xd, rad, nproj, 3, 8, ucols, 2.0)
    assert torch.equal(u1, u2)
    t = x.float() @ ra.float().t()
    ref = torch.zeros(M, ucols)
    for p in range(nproj):
        seg = t[:, p * 11:(p + 1) * 11]
        pr = torch.softmax(seg[:, :3], -1)
        for i in range(3):
            ref[:, p * 24 + i * 8:p * 24 + (i + 1) * 8] = 2.0 * pr[:, i:i + 1] * seg[:, 3:]
    _cmp(u1, ref, 7e-3, "route")
    assert (u1[:, nproj * 24:] == 0).all()


def test_hyperlora_route_in_a_workspace_sized_for_a_larger_chunk():
    """r06: one workspace sized by the query for the LARGEST chunk serves every smaller one.  The slice count rises as M falls (57 344 rows: 5 K slices,
    50 000 rows: 6), so slices * M is not monotone and the old query handed the last chunk of a merged prefill a buffer 5 % short (generate_avs_many
    with 512 samples: "hyperlora_route: workspace too small" from the layer sequencer, which cannot reallocate)."""
    from crab_amd import ops, _lib
    K, nproj, tcols, ucols = 4096, 3, 48, 96
    big, small = 57344, 50000
    ws = torch.empty((ops.hyperlora_route_workspace(big, K, tcols),), device="cuda", dtype=torch.uint8)
    assert ws.numel() >= 6 * small * tcols * 4                       # what the six-slice launch of the smaller call writes
    x = _rand(small, K, seed=11).cuda()
    ra = _rand(tcols, K, seed=12, scale=K ** -0.5)
    ra[nproj * 11:] = 0
    rad = ra.cuda()
    u_own = ops.hyperlora_route(x, rad, nproj, 3, 8, ucols, 2.0)
    u = torch.empty_like(u_own)
    lib = _lib.load()
    rc = lib.crab_hyperlora_route(_lib.ctx(0), None, x.data_ptr(), x.stride(0), rad.data_ptr(), rad.stride(0), small, K, nproj, 3, 8, u.data_ptr(), u.stride(0),
                                  ucols, 2.0, ws.data_ptr(), ws.numel())
    _lib.check(rc, 0)
    torch.cuda.synchronize()
    assert torch.equal(u, u_own)


@pytest.mark.parametrize("M", [1, 8, 17, 33, 64, 100, 128])
def test_gemm_skinny_regime(M):
    """M <= 128 dispatches to the weight-streaming kernel (all MT variants, N tail, K tail, second segment)."""
    from crab_amd import ops
    N, K, K2 = 1000 + 9, 1096, 32
    x, w, b, r = _rand(M, K, seed=3), _rand(N, K, seed=4, scale=K ** -0.5), _rand(N, seed=5), _rand(M, N, seed=6)
    x2, w2 = _rand(M, K2, seed=7), _rand(N, K2, seed=8, scale=0.1)
    y = ops.gemm(x.cuda(), w.cuda(), bias=b.cuda(), act="silu", residual=r.cuda(), x2=x2.cuda(), w2=w2.cuda(), out_fp32=True)
    z = F.silu(x.float() @ w.float().t() + x2.float() @ w2.float().t() + b.float()) + r.float()
    _cmp(y, z, TOL_F32, f"skinny M={M}")


@pytest.mark.parametrize("M,N,K,K2", [(16, 48, 64, 0), (13, 4096, 4096, 96), (5, 100, 8, 0), (8, 4096, 11008, 32), (1, 32017, 4096, 0), (16, 1000, 200, 32)])
def test_gemm_small_batch_lds_dma_kernel(M, N, K, K2):
    """M <= 16 (the reference's batch sizes): gemm_skinny_dma_kernel (weights through wave-private LDS-DMA rings) against fp32 arithmetic
    and against the register-direct kernel it replaces (tune 1): K shorter than one slot per wave (most waves idle), ragged K / N,
    the second K segment, the real projection shapes, bf16 and fp32 outputs, bias + residual."""
    from crab_amd import ops
    x, w = _rand(M, K, seed=31).cuda(), _rand(N, K, seed=32, scale=K ** -0.5).cuda()
    b, r = _rand(N, seed=33).cuda(), _rand(M, N, seed=34).cuda()
    x2 = _rand(M, K2, seed=35).cuda() if K2 else None
    w2 = _rand(N, K2, seed=36, scale=0.1).cuda() if K2 else None
    z = x.float() @ w.float().t() + (x2.float() @ w2.float().t() if K2 else 0) + b.float() + r.float()
    y = ops.gemm(x, w, bias=b, residual=r, x2=x2, w2=w2, out_fp32=True)
    _cmp(y, z.cpu(), TOL_F32, f"small-batch LDS-DMA kernel M={M} N={N} K={K}+{K2} (fp32 out)")
    y_old = ops.gemm(x, w, bias=b, residual=r, x2=x2, w2=w2, out_fp32=True, tune=1)
    _cmp(y, y_old.cpu(), TOL_F32, "LDS-DMA kernel vs register-direct kernel")
    yb = ops.gemm(x, w, bias=b, residual=r, x2=x2, w2=w2)
    assert torch.equal(yb, y.to(BF)), "bf16 output is not the rounded fp32 output"
    assert torch.equal(ops.gemm(x, w, bias=b, residual=r, x2=x2, w2=w2, out_fp32=True), y), "non-deterministic"


@pytest.mark.parametrize("M,tune", [(17, 0), (64, 0), (64, 104), (64, 208), (128, 0), (100, 103), (33, 102), (256, 403), (200, 407), (256, 0)])
def test_gemm_splitk_decode_regime(M, tune):
    """16 < M <= 256 with the caller workspace: split-K tiled kernels (tune 4xx: 256x256 ring kernel with K slices, ragged N / K
    tails / second segment) + fixed-order reduce epilogue."""
    from crab_amd import ops
    N, K, K2 = 1000 + 9, 1096, 32
    x, w, b, r = _rand(M, K, seed=3), _rand(N, K, seed=4, scale=K ** -0.5), _rand(N, seed=5), _rand(M, N, seed=6)
    x2, w2 = _rand(M, K2, seed=7), _rand(N, K2, seed=8, scale=0.1)
    args = dict(bias=b.cuda(), act="gelu", residual=r.cuda(), x2=x2.cuda(), w2=w2.cuda(), tune=tune)
    y = ops.gemm(x.cuda(), w.cuda(), out_fp32=True, **args)
    z = F.gelu(x.float() @ w.float().t() + x2.float() @ w2.float().t() + b.float()) + r.float()
    _cmp(y, z, TOL_F32, f"split-K M={M} tune={tune}")
    y2 = ops.gemm(x.cuda(), w.cuda(), out_fp32=True, **args)
    assert torch.equal(y, y2), "split-K reduction must be deterministic"
    yb = ops.gemm(x.cuda(), w.cuda(), **args)
    _cmp(yb, z, TOL_BF16, "bf16 out")


@pytest.mark.parametrize("M", [257, 300, 384, 511, 512])
def test_gemm_decode_panel_kernel_two_row_groups(M):
    """256 < M <= 512 (r04): the panel kernel over two 256-row groups in ONE launch (the two blocks of a weight panel side by side on one
    XCD).  Ragged N / K, second K segment, every epilogue input, fp32 and bf16 outputs, against fp32 arithmetic; and - the row groups run
    the very same block program - rows [0, 256) and [256, M) bit-identical to separate calls on those rows alone."""
    from crab_amd import ops
    N, K, K2 = 1000 + 9, 1096, 32
    x, w, b, r = _rand(M, K, seed=3), _rand(N, K, seed=4, scale=K ** -0.5), _rand(N, seed=5), _rand(M, N, seed=6)
    x2, w2 = _rand(M, K2, seed=7), _rand(N, K2, seed=8, scale=0.1)
    xd, wd, bd, rd, x2d, w2d = x.cuda(), w.cuda(), b.cuda(), r.cuda(), x2.cuda(), w2.cuda()
    y = ops.gemm(xd, wd, out_fp32=True, bias=bd, act="gelu", residual=rd, x2=x2d, w2=w2d)
    z = F.gelu(x.float() @ w.float().t() + x2.float() @ w2.float().t() + b.float()) + r.float()
    _cmp(y, z, TOL_F32, f"decode panel kernel, two row groups, M={M}")
    assert torch.equal(y, ops.gemm(xd, wd, out_fp32=True, bias=bd, act="gelu", residual=rd, x2=x2d, w2=w2d)), "must be deterministic"
    _cmp(ops.gemm(xd, wd, bias=bd, act="gelu", residual=rd, x2=x2d, w2=w2d), z, TOL_BF16, "two row groups, bf16 out")
    if M - 256 > 128:                      # the second group alone takes the panel kernel too (128 < rows <= 256): same program, same bits
        for lo, hi in ((0, 256), (256, M)):
            part = ops.gemm(xd[lo:hi], wd, out_fp32=True, bias=bd, act="gelu", residual=rd[lo:hi], x2=x2d[lo:hi], w2=w2d)
            assert torch.equal(y[lo:hi], part), f"rows {lo}:{hi} of the two-group launch differ from a launch on those rows alone"


@pytest.mark.parametrize("M", [65, 96, 128, 129, 200, 256])
@pytest.mark.parametrize("tune", [79601, 79602, 76401, 76404, 79605, 89602, 86404, 91601, 0])
def test_gemm_decode_panel_kernel(M, tune):
    """The decode panel kernels (64 < M <= 256 - r06: the row floor moved from 128 to 64, the row fragments beyond M are masked: the whole batch x a 96- or 64-wide weight panel per block; tune = 70000 + BN * 100 + K
    slices on the producer / consumer kernel, 80000 + ... on the 8-wave kernel, 91601 = 160-wide panels (projections wider than one round of
    96-wide panels), 0 = automatic choice): ragged N (last panel partly outside),
    K not a multiple of the 64-wide slot, a second K segment, every epilogue input (bias, GELU, residual), fp32 and bf16 outputs;
    deterministic; the two kernels issue the same MFMAs in the same order: bit-identical."""
    from crab_amd import ops
    N, K, K2 = 1000 + 9, 1096, 32
    x, w, b, r = _rand(M, K, seed=3), _rand(N, K, seed=4, scale=K ** -0.5), _rand(N, seed=5), _rand(M, N, seed=6)
    x2, w2 = _rand(M, K2, seed=7), _rand(N, K2, seed=8, scale=0.1)
    args = dict(bias=b.cuda(), act="gelu", residual=r.cuda(), x2=x2.cuda(), w2=w2.cuda(), tune=tune)
    y = ops.gemm(x.cuda(), w.cuda(), out_fp32=True, **args)
    z = F.gelu(x.float() @ w.float().t() + x2.float() @ w2.float().t() + b.float()) + r.float()
    _cmp(y, z, TOL_F32, f"decode panel kernel M={M} tune={tune}")
    assert torch.equal(y, ops.gemm(x.cuda(), w.cuda(), out_fp32=True, **args)), "must be deterministic"
    _cmp(ops.gemm(x.cuda(), w.cuda(), **args), z, TOL_BF16, "decode panel kernel, bf16 out")
    if 70000 <= tune < 80000:
        args["tune"] = tune + 10000
        assert torch.equal(y, ops.gemm(x.cuda(), w.cuda(), out_fp32=True, **args)), "producer / consumer kernel != 8-wave kernel"
    if tune == 91601:                           # same K order, same MFMAs per output element as the 96-wide panels with one K slice
        args["tune"] = 79601
        assert torch.equal(y, ops.gemm(x.cuda(), w.cuda(), out_fp32=True, **args)), "160-wide panels != 96-wide panels"


@pytest.mark.parametrize("name,N,K,K2", [("qkv", 12288, 4096, 96), ("o", 4096, 4096, 32), ("gate|up", 22016, 4096, 64), ("down", 4096, 11008, 32),
                                          ("qwen qkv", 4608, 3584, 96), ("qwen gate|up", 37888, 3584, 64), ("qwen down", 3584, 18944, 32)])
@pytest.mark.parametrize("M", [80, 128, 256, 448])
def test_gemm_decode_panel_kernel_projection_shapes(name, N, K, K2, M):
    """The automatic decomposition on the real decoder projections at M = 256 and M = 448 (two row groups) (Llama-2-7B and Qwen2-7B widths)
    against fp32 arithmetic, and against the older split-K kernels (tune 104: 128x128 tiles, 4 slices) on the same operands."""
    from crab_amd import ops
    g = torch.Generator(device="cuda").manual_seed(N + K)
    x = torch.randn(M, K, device="cuda", generator=g).to(BF)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(BF)
    x2 = torch.randn(M, K2, device="cuda", generator=g).to(BF)
    w2 = (torch.randn(N, K2, device="cuda", generator=g) * 0.1).to(BF)
    y = ops.gemm(x, w, x2=x2, w2=w2, out_fp32=True)
    z = x.float() @ w.float().t() + x2.float() @ w2.float().t()
    _cmp(y, z.cpu(), TOL_F32 * 4, f"decode panel kernel, {name} at M={M} (vs torch fp32 matmul on the GPU)")
    y1 = ops.gemm(x[:256], w, x2=x2[:256], w2=w2, out_fp32=True, tune=104)
    _cmp(y[:256], y1.cpu(), TOL_F32, f"decode panel kernel vs 128x128 split-K kernel, {name}")
    with ops.launch_trace() as tr:
        ops.gemm(x, w, x2=x2, w2=w2, out_fp32=True)
    assert tr.launched("gemm_dec2_kernel" if M > 256 else "gemm_dec_ws_kernel") + tr.launched("gemm_dec_kernel<160>") == 1, tr.counts


@pytest.mark.parametrize("res_fp32", [True, False])
@pytest.mark.parametrize("M,N", [(64, 4096), (8, 512), (40, 1024), (300, 512), (256, 4096), (2048, 1024)])
def test_gemm_fused_post_rmsnorm(M, N, res_fp32):
    """C = x W^T + R and norm_out = rmsnorm(C)*w: fused split-K epilogue (16 < M <= 256), unfused elsewhere; with the residual stream in
    fp32 (R == C fp32: stored unrounded, the norm reads the fp32 row, one rounding of the normalised row) and in bf16 (r01-r03 storage)."""
    from crab_amd import ops
    from oracle import crab_oracle as O
    K = 1024
    x, w, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(M, N, seed=3)
    nw = (1 + 0.1 * torch.randn(N)).to(BF)
    xd = r.cuda().float() if res_fp32 else r.cuda().clone()      # in-place residual: out == residual buffer
    h = torch.empty(M, N, dtype=BF, device="cuda")
    ops.gemm(x.cuda(), w.cuda(), residual=xd, out=xd, post_norm=(nw.cuda(), 1e-5, h))
    c_ref = (x.float() @ w.float().t() + r.float())
    _cmp(xd, c_ref, TOL_F32 * 4 if res_fp32 else TOL_BF16, f"C, {'fp32' if res_fp32 else 'bf16'} residual stream M={M} N={N}")
    with O.residual_storage(res_fp32):
        _cmp(h, O.rmsnorm(xd.cpu().float(), nw.float(), 1e-5, emulate=BF), 4.5e-3,
             f"post-norm, {'fp32' if res_fp32 else 'bf16'} residual stream M={M} N={N}")


@pytest.mark.parametrize("res_fp32", [True, False])
@pytest.mark.parametrize("M", [1, 3, 8, 16])
@pytest.mark.parametrize("N,K,nproj_next", [(4096, 4096, 2), (4096, 11008, 3), (128, 64, 3), (3584, 2048, 0), (200, 72, 1)])
def test_small_batch_layer_tail_rowfin(M, N, K, nproj_next, res_fp32):
    """csrc/rowfin.hip, the M <= 16 tail behind o_proj / down_proj (modeling_llama.py:805-827, lora.py:338-350): the projection's own
    router rows ride on the GEMM launch, then two wide launches apply the hyper-LoRA update, store the residual row, its RMSNorm and the
    NEXT group's router mix.  Against fp32 arithmetic on the same bf16 inputs, and against the K-extension form (router launches +
    second K segment) the larger batches use; run twice (the arrival counter must come back to zero)."""
    from crab_amd import ops
    from oracle import crab_oracle as O
    nl, r, sc = 3, 8, 2.0
    x, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    res = _rand(M, N, seed=3)
    bias = _rand(N, seed=4, scale=0.1) if N == 128 else None
    RA = torch.zeros(16, K, dtype=BF)
    RA[:nl + r] = _rand(nl + r, K, seed=5, scale=K ** -0.5)
    B2 = torch.zeros(N, 32, dtype=BF)
    B2[:, :nl * r] = _rand(N, nl * r, seed=6, scale=0.2)
    nw = (1 + 0.1 * torch.randn(N, generator=torch.Generator().manual_seed(7))).to(BF)
    route = None
    if nproj_next:
        tc = (nproj_next * (nl + r) + 15) // 16 * 16
        uc = (nproj_next * nl * r + 31) // 32 * 32
        RAn = torch.zeros(tc, N, dtype=BF)
        RAn[:nproj_next * (nl + r)] = _rand(nproj_next * (nl + r), N, seed=8, scale=N ** -0.5)
        un = torch.full((M, uc), float("nan"), dtype=BF, device="cuda")
        route = (RAn.cuda(), nproj_next, nl, r, uc, sc, un)
    xd, wd, RAd, B2d, nwd = x.cuda(), w.cuda(), RA.cuda(), B2.cuda(), nw.cuda()
    outs = []
    rdt = torch.float32 if res_fp32 else BF      # the residual stream's storage (fp32 since r04; bf16 = the r01-r03 form, still in the library)
    for rep in range(2):
        c = res.cuda().to(rdt)
        h = torch.empty(M, N, dtype=BF, device="cuda")
        ops.gemm(xd, wd, bias=bias.cuda() if bias is not None else None, residual=c, out=c, post_norm=(nwd, 1e-5, h), route=route,
                 lora_self=(RAd, nl, r, sc, B2d))
        outs.append((c.clone(), h.clone(), route[6].clone() if route else None))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and (route is None or torch.equal(outs[0][2], outs[1][2]))
    c, h, un_got = outs[0]
    # fp32 reference
    t = x.float() @ RA[:nl + r].float().t()
    p = torch.softmax(t[:, :nl], -1)
    u = (sc * p[:, :, None] * t[:, None, nl:]).reshape(M, nl * r).to(BF).float()
    y = x.float() @ w.float().t() + (bias.float() if bias is not None else 0) + res.float() + u @ B2[:, :nl * r].float().t()
    _cmp(c, y, 2e-4 if res_fp32 else TOL_BF16, f"rowfin: residual row with deferred hyper-LoRA update M={M} N={N} K={K} {'fp32' if res_fp32 else 'bf16'} stream")
    with O.residual_storage(res_fp32):
        _cmp(h, O.rmsnorm(c.cpu().float(), nw.float(), 1e-5, emulate=BF), 4.5e-3 if res_fp32 else 2e-3, "rowfin: post-norm row")
    if route:
        tn = h.cpu().float() @ route[0].cpu().float().t()
        refu = torch.zeros(M, route[4])
        for pj in range(nproj_next):
            tt = tn[:, pj * (nl + r):(pj + 1) * (nl + r)]
            pp = torch.softmax(tt[:, :nl], -1)
            refu[:, pj * nl * r:(pj + 1) * nl * r] = (sc * pp[:, :, None] * tt[:, None, nl:]).reshape(M, nl * r)
        _cmp(un_got, refu, TOL_BF16, "rowfin: next group's router mix")
        assert float(un_got[:, nproj_next * nl * r:].float().abs().max()) == 0.0 if route[4] > nproj_next * nl * r else True
    # the K-extension form (what M > 16 runs): router launches + second K segment
    ud = ops.hyperlora_route(xd, RAd, 1, nl, r, 32, sc)
    c2 = res.cuda().to(rdt)
    h2 = torch.empty(M, N, dtype=BF, device="cuda")
    ops.gemm(xd, wd, bias=bias.cuda() if bias is not None else None, residual=c2, x2=ud, w2=B2d, out=c2, post_norm=(nwd, 1e-5, h2))
    _cmp(c, c2.float(), 2e-4 if res_fp32 else TOL_BF16, "rowfin vs K-extension form: residual row (HIP vs HIP)")


@pytest.mark.parametrize("hd,B,H,Hk,pos", [(128, 1, 32, 32, 702), (128, 1, 32, 32, 5), (128, 8, 32, 32, 830), (128, 2, 28, 4, 333), (64, 3, 4, 2, 0),
                                           (64, 1, 4, 4, 77), (128, 16, 32, 32, 100),
                                           # empty splits (8 splits, fewer keys), the first token, more than 128 keys per split, the last cache row
                                           (128, 1, 4, 4, 0), (128, 1, 4, 4, 3), (128, 4, 32, 32, 1000), (128, 2, 8, 8, 1022)])
def test_decode_attention_fused_rope_append_split(hd, B, H, Hk, pos):
    """crab_attn_decode_rope vs crab_qkv_rope_split + crab_attn_decode: the cache rows it appends are BIT-identical (same rotation,
    same rounding), the attention output agrees to the accumulation order (new key last, splits merged in order), for one block per
    (b, h) and for the context split over up to 8 blocks (B * H < 512), GQA included; two consecutive calls (tickets back to zero)."""
    from crab_amd import ops
    Tmax, theta = 1024, 10000.0
    tab = ops.rope_table(Tmax, hd, theta, "cuda")
    g = torch.Generator(device="cuda").manual_seed(pos + B)
    kc = (torch.randn(B, Hk, Tmax, hd, device="cuda", generator=g) * 0.5).to(BF)
    vc = (torch.randn(B, Hk, Tmax, hd, device="cuda", generator=g) * 0.5).to(BF)
    ws = ops.attn_decode_rope_workspace(B, H, hd, "cuda")
    pd = torch.tensor([pos], dtype=torch.int32, device="cuda")
    for step in range(2):
        qkv = (torch.randn(B, (H + 2 * Hk) * hd, device="cuda", generator=g)).to(BF)
        k1, v1, k2, v2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
        q1 = qkv.clone()
        ops.qkv_rope_split(q1, tab, k1, v1, None, B, 1, H, Hk, hd, Tmax, pos0=0, pos_dev=pd)
        o1 = torch.zeros(B, H * hd, dtype=BF, device="cuda")
        ops.attn_decode(q1, k1, v1, o1, B, H, Hk, hd, Tmax, 1, hd ** -0.5, ctx_dev=pd)
        o2 = torch.zeros(B, H * hd, dtype=BF, device="cuda")
        ops.attn_decode_rope(qkv, tab, k2, v2, o2, B, H, Hk, hd, Tmax, 0, hd ** -0.5, pos_dev=pd, workspace=ws)
        assert torch.equal(k1, k2) and torch.equal(v1, v2), "appended cache rows differ from the unfused pair"
        # q / k take the same roundings as the unfused pair; the softmax accumulation order differs (the new key last, the split merge)
        _cmp(o2, o1.float(), 6e-3, f"fused rope + append + split-context decode attention vs the unfused pair (HIP vs HIP) B={B} H={H} pos={pos}")
        kc, vc = k2, v2
        pd += 1
    assert int(ws[-B * H * 4:].view(torch.int32).abs().sum()) == 0


def test_rowfin_in_call_lora_is_refused_outside_its_regime():
    from crab_amd import ops
    from crab_amd._lib import CrabHipError
    M, N, K = 32, 128, 64
    x, w, RA, B2 = _rand(M, K).cuda(), _rand(N, K).cuda(), _rand(16, K).cuda(), _rand(N, 32).cuda()
    h = torch.empty(M, N, dtype=BF, device="cuda")
    with pytest.raises(CrabHipError, match="M <= 16"):
        ops.gemm(x, w, post_norm=(torch.ones(N, dtype=BF, device="cuda"), 1e-5, h), lora_self=(RA, 3, 8, 2.0, B2))


def test_gemm_ring_split_wide_projection_auto():
    """N >= 10240 at M = 256 takes the 256x256 ring kernel with K slices automatically; result == the 128x128 split path."""
    from crab_amd import ops
    M, N, K, K2 = 256, 10240 + 256, 1024, 32
    x, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    x2, w2 = _rand(M, K2, seed=3), _rand(N, K2, seed=4, scale=0.1)
    y0 = ops.gemm(x.cuda(), w.cuda(), x2=x2.cuda(), w2=w2.cuda(), out_fp32=True)                 # automatic: ring, 6 tiles -> 5 slices
    y1 = ops.gemm(x.cuda(), w.cuda(), x2=x2.cuda(), w2=w2.cuda(), out_fp32=True, tune=104)       # 128x128 kernel, 4 slices
    z = x.float() @ w.float().t() + x2.float() @ w2.float().t()
    _cmp(y0, z, TOL_F32, "ring split auto")
    assert (y0 - y1).abs().max().item() < 1e-3 * z.abs().max().item()


def test_decode_wide_projections_ring_split_with_fused_epilogues():
    """The two decode GEMMs that take the 256x256 ring kernel with K slices at M = 256, with their fused reductions:
    q|k|v (N = 12288, 5 slices) + RoPE / KV append == unfused pair bit for bit; gate|up (N = 22016, 2 slices) + SwiGLU."""
    from crab_amd import ops
    M, K, K2, H, d, Tmax = 256, 512, 32, 32, 128, 64
    x, x2 = _rand(M, K, seed=1), _rand(M, K2, seed=3)
    # q|k|v
    N = 3 * H * d
    w, w2 = _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, K2, seed=4, scale=0.1)
    tab = ops.rope_table(Tmax, d, 10000.0, "cuda")
    pos = torch.tensor([9], dtype=torch.int32, device="cuda")
    outs = []
    for fused in (False, True):
        kc = torch.zeros(M, H, Tmax, d, dtype=BF, device="cuda")
        vc = torch.zeros_like(kc)
        y = ops.gemm(x.cuda(), w.cuda(), x2=x2.cuda(), w2=w2.cuda(), rope=(tab, kc, vc, H, H, d, Tmax, 0, pos) if fused else None)
        if not fused:
            ops.qkv_rope_split(y, tab, kc, vc, None, M, 1, H, H, d, Tmax, pos0=0, pos_dev=pos)
        outs.append((y[:, :H * d].clone(), kc[:, :, 9].clone(), vc[:, :, 9].clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    z = x.float() @ w.float().t() + x2.float() @ w2.float().t()
    _cmp(outs[1][2].reshape(M, H * d), z[:, 2 * H * d:], TOL_BF16, "v rows in the cache")
    # gate|up
    I = 11008
    wg, wu = _rand(I, K, seed=5, scale=K ** -0.5), _rand(I, K, seed=6, scale=K ** -0.5)
    wi = torch.stack([wg, wu], 1).reshape(2 * I, K).contiguous()
    y = ops.gemm(x.cuda(), wi.cuda(), act="swiglu_pair")
    _cmp(y, F.silu(x.float() @ wg.float().t()) * (x.float() @ wu.float().t()), TOL_BF16, "gate|up ring split + swiglu")


@pytest.mark.parametrize("M", [4, 48, 256])
@pytest.mark.parametrize("nproj", [3, 2])
def test_gemm_post_norm_routes_next_group(M, nproj):
    """o_proj / down_proj GEMM with the fused post-RMSNorm that also evaluates the NEXT group's hyper-LoRA router on the
    normalised rows (row-owning split-K reduction; separate pass on the skinny path) == norm followed by hyperlora_route."""
    from crab_amd import ops
    K, N = 1024, 2048
    tcols, ucols = (nproj * 11 + 15) // 16 * 16, (nproj * 24 + 31) // 32 * 32
    x, w, r, nw = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(M, N, seed=3), (1 + 0.1 * _rand(N, seed=4).float()).to(BF)
    ra = _rand(tcols, N, seed=5, scale=N ** -0.5)
    ra[nproj * 11:] = 0
    rd = r.cuda().clone()
    h = torch.empty(M, N, dtype=BF, device="cuda")
    u = torch.full((M, ucols), 7.0, dtype=BF, device="cuda")
    ops.gemm(x.cuda(), w.cuda(), residual=rd, out=rd, post_norm=(nw.cuda(), 1e-5, h), route=(ra.cuda(), nproj, 3, 8, ucols, 2.0, u))
    h2 = torch.empty_like(h)
    r2 = r.cuda().clone()
    ops.gemm(x.cuda(), w.cuda(), residual=r2, out=r2, post_norm=(nw.cuda(), 1e-5, h2))
    assert torch.equal(h, h2) and torch.equal(rd, r2)
    u_ref = ops.hyperlora_route(h2, ra.cuda(), nproj, 3, 8, ucols, 2.0)
    # M > 16: the row-owning reduction routes on the stored bf16 h, like the stand-alone router: identical.  M <= 16 (the small-batch tail, r04): the
    # router partials are formed on the UNROUNDED row times the norm weight and scaled by rstd afterwards (rowfin.hip: the second launch then waits for
    # nothing) - one bf16 rounding less than the reference-shaped order, so u agrees to a bf16 ulp of its largest entry, not bit for bit
    _cmp(u, u_ref.float().cpu(), 1e-6 if M > 16 else 8e-3, "route ahead")
    if M <= 16:
        t = (h2.float() @ ra.cuda().float().t()).cpu()
        hx = (rd.float() * torch.rsqrt(rd.float().pow(2).mean(-1, keepdim=True) + 1e-5) * nw.cuda().float())
        tx = (hx @ ra.cuda().float().t()).cpu()                       # router inputs without the bf16 rounding of h: what the tail now computes
        ux = torch.zeros(M, ucols)
        for p_ in range(nproj):
            tt = tx[:, p_ * 11:(p_ + 1) * 11]
            ux[:, p_ * 24:(p_ + 1) * 24] = (2.0 * torch.softmax(tt[:, :3], -1)[:, :, None] * tt[:, None, 3:]).reshape(M, 24)
        _cmp(u, ux, 4.5e-3, "route ahead vs the unrounded router (fp32 math, bf16 output)")
    assert (u[:, nproj * 24:] == 0).all()


@pytest.mark.parametrize("res_fp32", [False, True])
def test_router_ahead_is_batch_invariant_to_a_bf16_ulp_across_the_m16_boundary(res_fp32):
    """the M <= 16 layer tail (rowfin.hip) forms the NEXT group's router logits on the unrounded product (y * w) * rstd, every
    M > 16 path routes on the stored bf16 row h - so a clip's router input is not bit-identical between a batch of 16 and a batch of 17.
    Stated in DESIGN.md 3 (kernel table, rowfin) and bounded here: the same 16 rows through both paths give u within one bf16 ulp of its
    largest entry, and the stored rows (x, h) are identical."""
    from crab_amd import ops
    K, N, nproj = 1024, 2048, 3
    tcols, ucols = 48, 96
    x, w, nw = _rand(17, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), (1 + 0.1 * _rand(N, seed=4).float()).to(BF)
    r = _rand(17, N, seed=3)
    ra = _rand(tcols, N, seed=5, scale=N ** -0.5)
    ra[nproj * 11:] = 0
    outs = []
    for M in (16, 17):
        rd = (r[:M].float() if res_fp32 else r[:M]).cuda().clone()
        h = torch.empty(M, N, dtype=BF, device="cuda")
        u = torch.zeros((M, ucols), dtype=BF, device="cuda")
        ops.gemm(x[:M].cuda(), w.cuda(), residual=rd, out=rd, post_norm=(nw.cuda(), 1e-5, h), route=(ra.cuda(), nproj, 3, 8, ucols, 2.0, u))
        outs.append((rd[:16].float().cpu(), h[:16].float().cpu(), u[:16].float().cpu()))
    (x16, h16, u16), (x17, h17, u17) = outs
    _cmp(x16, x17, 1.2e-5 if res_fp32 else 1e-6 + 4e-3, "residual row, M = 16 tail vs M = 17 reduction")      # different K splits: fp32 summation order (+ one bf16 rounding when stored in bf16)
    _cmp(h16, h17, 4.5e-3, "normalised row, M = 16 tail vs M = 17 reduction")
    _cmp(u16, u17, 8e-3, "router output u of the same rows at M = 16 (unrounded router input) vs M = 17 (stored bf16 row)")


@pytest.mark.parametrize("M", [1, 3, 8, 16, 40, 256])
@pytest.mark.parametrize("H,Hk,bias,d", [(4, 4, False, 128), (8, 2, True, 128), (4, 2, True, 64)])
def test_gemm_fused_rope_kv_append_equals_unfused_pair(M, H, Hk, bias, d):
    """q|k|v projection with the fused RoPE + KV-cache append (decode: one row per sequence) must leave exactly what the
    projection followed by qkv_rope_split(S = 1) leaves: M <= 16 in the epilogue of the small-batch kernel (a block owns both halves of
    its rotation pairs), larger M in the split-K reduction kernel."""
    from crab_amd import ops
    K, Tmax, K2 = 1024, 64, 32
    N = (H + 2 * Hk) * d
    x, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5)
    x2, w2 = _rand(M, K2, seed=3), _rand(N, K2, seed=4, scale=0.1)
    b = _rand(N, seed=5).cuda() if bias else None
    tab = ops.rope_table(Tmax, d, 10000.0, "cuda")
    pos = torch.tensor([17], dtype=torch.int32, device="cuda")
    outs = []
    for fused in (False, True):
        kc = torch.zeros(M, Hk, Tmax, d, dtype=BF, device="cuda")
        vc = torch.zeros_like(kc)
        rope = (tab, kc, vc, H, Hk, d, Tmax, 0, pos)
        y = ops.gemm(x.cuda(), w.cuda(), bias=b, x2=x2.cuda(), w2=w2.cuda(), rope=rope if fused else None)
        if not fused:
            ops.qkv_rope_split(y, tab, kc, vc, None, M, 1, H, Hk, d, Tmax, pos0=0, pos_dev=pos)
        outs.append((y[:, :H * d].clone(), kc, vc))
    for a, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a, b_)
    assert outs[1][1][:, :, 17].abs().sum() > 0 and outs[1][1][:, :, 16].abs().sum() == 0


@pytest.mark.parametrize("M,tune", [(5, 0), (16, 0), (64, 0), (200, 0), (256, 0), (700, 0), (1500, 302), (1500, 301), (1500, 300)])
def test_gemm_swiglu_pair_epilogue(M, tune):
    """Interleaved (gate_i, up_i) weight rows + SwiGLU in the GEMM epilogue == silu(x Wg^T) * (x Wu^T) in every kernel
    regime (skinny, split-K + reduction kernel, 128^2 two-stage, 256^2 ring, register-staged), with the LoRA K segment."""
    from crab_amd import ops
    K, I, K2 = 512, 1376, 32
    x, wg, wu = _rand(M, K, seed=1), _rand(I, K, seed=2, scale=K ** -0.5), _rand(I, K, seed=3, scale=K ** -0.5)
    x2, bg, bu = _rand(M, K2, seed=4), _rand(I, K2, seed=5, scale=0.1), _rand(I, K2, seed=6, scale=0.1)
    w = torch.stack([wg, wu], 1).reshape(2 * I, K).contiguous()
    w2 = torch.stack([bg, bu], 1).reshape(2 * I, K2).contiguous()
    y = ops.gemm(x.cuda(), w.cuda(), x2=x2.cuda(), w2=w2.cuda(), act="swiglu_pair", tune=tune)
    assert y.shape == (M, I)
    g = x.float() @ wg.float().t() + x2.float() @ bg.float().t()
    u = x.float() @ wu.float().t() + x2.float() @ bu.float().t()
    _cmp(y, F.silu(g) * u, TOL_BF16, "swiglu pair")
    y32 = ops.gemm(x.cuda(), w.cuda(), x2=x2.cuda(), w2=w2.cuda(), act="swiglu_pair", out_fp32=True, tune=tune)
    _cmp(y32, F.silu(g) * u, TOL_F32, "swiglu pair fp32")
    with pytest.raises(Exception):
        ops.gemm(x.cuda(), w.cuda(), act="swiglu_pair", residual=y)          # no residual with the pair epilogue


@pytest.mark.parametrize("tune", [302])
@pytest.mark.parametrize("M,N,K,K2", [(2808, 4096, 1024, 96), (1100, 1300, 520, 0), (5616, 2048, 256, 32), (6000, 11000, 72, 0),
                                      (300, 256, 4096, 0)])
def test_gemm_big_ring_kernel(M, N, K, K2, tune):
    """256x256 ring kernel (forced with tune=302; (6000, 11000): 1032 tiles = 4.03 rounds of 256 blocks with 3 K tiles each,
    (300, 256): two tiles): ragged M/N tiles, K tails inside a 32-wide stage, second segment."""
    from crab_amd import ops
    x, w, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=K ** -0.5), _rand(N, seed=3), _rand(M, N, seed=4)
    x2 = _rand(M, K2, seed=5) if K2 else None
    w2 = _rand(N, K2, seed=6, scale=0.1) if K2 else None
    y = ops.gemm(x.cuda(), w.cuda(), bias=b.cuda(), act="gelu", residual=r.cuda(), x2=x2.cuda() if K2 else None,
                 w2=w2.cuda() if K2 else None, out_fp32=True, tune=tune)
    z = x.float() @ w.float().t() + b.float()
    if K2:
        z = z + x2.float() @ w2.float().t()
    _cmp(y, F.gelu(z) + r.float(), TOL_F32, "ring 256")
    y2 = ops.gemm(x.cuda(), w.cuda(), bias=b.cuda(), act="gelu", residual=r.cuda(), x2=x2.cuda() if K2 else None,
                  w2=w2.cuda() if K2 else None, out_fp32=True, tune=301)
    assert (y - y2).abs().max().item() < 1e-3 * z.abs().max().item(), "ring vs 2-stage kernel disagree"


def test_invalid_arguments_fail_loudly_with_a_message():
    """C-ABI error behaviour: bad shapes / alignments / unsupported variants return a negative code with a
    message in crab_last_error, surfaced as CrabHipError by the Python side; nothing falls back silently."""
    from crab_amd import ops
    from crab_amd._lib import CrabHipError
    x = _rand(8, 20, seed=1).cuda()                     # K = 20 is not a multiple of 8
    with pytest.raises(CrabHipError, match="multiples of 8"):
        ops.gemm(x, _rand(16, 20, seed=2).cuda())
    a, w = _rand(8, 64, seed=1).cuda(), _rand(16, 64, seed=2).cuda()
    with pytest.raises(CrabHipError, match="A2/B2"):
        from crab_amd._lib import GemmDesc
        g = GemmDesc()
        out = torch.empty(8, 16, dtype=BF, device="cuda")
        g.A, g.B, g.C, g.A2 = a.data_ptr(), w.data_ptr(), out.data_ptr(), a.data_ptr()
        g.lda = g.ldb = 64; g.ldc = 16; g.M, g.N, g.K = 8, 16, 64; g.batch = g.nb0 = 1; g.res_scale = 1.0
        ops.gemm_desc(g)
    with pytest.raises(CrabHipError, match="swiglu-pair"):
        ops.gemm(a, _rand(18, 64, seed=3).cuda(), act="swiglu_pair")            # N % 4 != 0
    q = _rand(2, 4 * 96, seed=4).cuda()
    kc = torch.zeros(2, 4, 16, 96, dtype=BF, device="cuda")
    with pytest.raises(CrabHipError, match="head_dim"):
        ops.attn_decode(q, kc, kc, torch.empty_like(q), 2, 4, 4, 96, 16, 4, 0.1)
    with pytest.raises(CrabHipError, match="overflow"):
        ops.qkv_rope_split(_rand(2 * 5, 3 * 4 * 64, seed=5).cuda(), ops.rope_table(16, 64, 1e4, "cuda"), torch.zeros(2, 4, 4, 64, dtype=BF, device="cuda"),
                           torch.zeros(2, 4, 4, 64, dtype=BF, device="cuda"), None, 2, 5, 4, 4, 64, 4, pos0=0)


@pytest.mark.parametrize("with_vt", [True, False])
@pytest.mark.parametrize("B,S,H,Hk,bias,ids", [(5, 300, 32, 32, False, False), (3, 702, 28, 4, True, False), (4, 333, 16, 16, False, True), (2, 1000, 32, 32, False, False)])
def test_prefill_qkv_projection_rotates_in_its_epilogue(B, S, H, Hk, bias, ids, with_vt):
    """Prefill q|k|v projection with the RoPE of q / k and the K-cache append in the GEMM epilogue (crab_gemm_desc.rope_S) followed by the
    v-only split == projection followed by the full qkv_rope_split, bit for bit: q columns of C, K cache, V cache, V^T; ragged last row tile,
    rows of several sequences in one tile, grouped kv heads + bias (Qwen2), explicit rotary positions (forward()'s position_ids)."""
    from crab_amd import ops
    d, K, Tmax, pos0 = 128, 1024, 1024, 5
    M, N = B * S, (H + 2 * Hk) * d
    g = torch.Generator(device="cuda").manual_seed(B * S)
    x = torch.randn(M, K, device="cuda", generator=g).to(BF)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(BF)
    bv = torch.randn(N, device="cuda", generator=g).to(BF) if bias else None
    tab = ops.rope_table(2048, d, 10000.0, "cuda")
    pid = (torch.randint(0, 2000, (B, S), device="cuda", generator=g).to(torch.int32)) if ids else None
    Sp = (S + 7) // 8 * 8
    outs = []
    for fused in (True, False):
        kc = torch.zeros(B, Hk, Tmax, d, dtype=BF, device="cuda")
        vc = torch.zeros_like(kc)
        vt = torch.zeros(B, Hk, d, Sp, dtype=BF, device="cuda")
        qkv = torch.empty(M, N, dtype=BF, device="cuda")
        if fused:
            info = {}
            ops.gemm(x, w, bias=bv, out=qkv, rope=(tab, kc, vc, H, Hk, d, Tmax, pos0, None, S, pid, vt if with_vt else None), info=info)
            assert info["fused_prefill_rope"] == (2 if with_vt else 1), "the library declined the fused prefill RoPE at a shape it is built for"
            if not with_vt:
                ops.qkv_rope_split(qkv, None, None, vc, vt, B, S, H, Hk, d, Tmax, pos0=pos0)
        else:
            ops.gemm(x, w, bias=bv, out=qkv)
            ops.qkv_rope_split(qkv, tab, kc, vc, vt, B, S, H, Hk, d, Tmax, pos0=pos0, pos_ids=pid)
        outs.append((qkv[:, :H * d].clone(), kc, vc, vt))
    # V^T beyond a sequence's last token (the pad of the last octet) is zero in both forms (the unfused pass writes zeros, the fused one packs them)
    for name, a, b in zip(("q", "k cache", "v cache", "v^T"), outs[0], outs[1]):
        assert torch.equal(a, b), f"fused prefill RoPE: {name} differs from the unfused pair"
    assert float(outs[0][1][:, :, pos0:pos0 + S].float().abs().sum()) > 0


def test_prefill_rope_form_with_one_row_per_sequence_is_not_read_as_the_decode_form():
    """ops.gemm's PREFILL rope tuple with S == 1 (a one-token prompt, or a chunk of one row at a cache offset): the library reads rope_S <= 1 as the
    decode form and answers 0 to "do you fuse the prefill rotation" - the caller's split pass would then rotate q / k a second time (found by
    scripts/fuzz_rope_epilogue.py at pos0 > 0; at position 0 the second rotation is the identity, which is why no model test saw it).  The wrapper
    hands such a call to the library without rope fields."""
    from crab_amd import ops
    B, S, H, Hk, d, K, Tmax, pos0 = 3, 1, 4, 2, 128, 256, 64, 9
    N = (H + 2 * Hk) * d
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(B * S, K, device="cuda", generator=g).to(BF)
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(BF)
    tab = ops.rope_table(Tmax, d, 10000.0, "cuda")
    outs = []
    for form in ("prefill-tuple", "plain"):
        kc = torch.zeros(B, Hk, Tmax, d, dtype=BF, device="cuda"); vc = torch.zeros_like(kc)
        vt = torch.zeros(B, Hk, d, 8, dtype=BF, device="cuda")
        qkv = torch.empty(B * S, N, dtype=BF, device="cuda")
        if form == "plain":
            ops.gemm(x, w, out=qkv)
        else:
            info = {}
            ops.gemm(x, w, out=qkv, rope=(tab, kc, vc, H, Hk, d, Tmax, pos0, None, S, None, vt), info=info)
            assert info["fused_prefill_rope"] == 0
        ops.qkv_rope_split(qkv, tab, kc, vc, vt, B, S, H, Hk, d, Tmax, pos0=pos0)
        outs.append((qkv[:, :H * d].clone(), kc, vc))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M", [1, 8, 16, 40, 256, 448])
@pytest.mark.parametrize("H,Hk,bias,d", [(4, 4, False, 128), (8, 2, True, 128), (4, 2, True, 64)])
def test_gemm_fused_rope_ragged_rows_rotate_at_their_own_positions(M, H, Hk, bias, d):
    """The ragged decode batch (crab_gemm_desc.rope_row_off, ABI 9): row m of the fused q|k|v projection is rotated at slot - row_off[m] while
    its K / V rows land in the common slot.  Bit-identical, row by row, to the plain fused call issued at position slot - row_off[m] - in the
    M <= 16 epilogue (skinny.hip), the split-K reduction (gemm.hip, incl. the two-row-group regime above 256 rows) and the stand-alone pass
    (crab_qkv_rope_split_ragged)."""
    from crab_amd import ops
    K, Tmax, K2, slot = 1024, 64, 32, 29
    N = (H + 2 * Hk) * d
    x, w = _rand(M, K, seed=1).cuda(), _rand(N, K, seed=2, scale=K ** -0.5).cuda()
    x2, w2 = _rand(M, K2, seed=3).cuda(), _rand(N, K2, seed=4, scale=0.1).cuda()
    b = _rand(N, seed=5).cuda() if bias else None
    tab = ops.rope_table(Tmax, d, 10000.0, "cuda")
    offs = [0, 3, 11, 29]
    off = torch.tensor([offs[(5 * m + m // 3) % 4] for m in range(M)], dtype=torch.int32)
    pos = torch.tensor([slot], dtype=torch.int32, device="cuda")

    def run(p, row_off, split_pass):
        kc = torch.zeros(M, Hk, Tmax, d, dtype=BF, device="cuda")
        vc = torch.zeros_like(kc)
        pd = torch.tensor([p], dtype=torch.int32, device="cuda")
        if split_pass:
            y = ops.gemm(x, w, bias=b, x2=x2, w2=w2)
            ops.qkv_rope_split(y, tab, kc, vc, None, M, 1, H, Hk, d, Tmax, pos0=0, pos_dev=pd, row_off=row_off)
        else:
            y = ops.gemm(x, w, bias=b, x2=x2, w2=w2, rope=(tab, kc, vc, H, Hk, d, Tmax, 0, pd), rope_row_off=row_off)
        return y[:, :H * d].clone(), kc, vc

    for split_pass in (False, True):
        # the reference of each form is the SAME form without offsets (above 256 rows the projection's K split depends on whether the RoPE is
        # fused behind it, so the fused and the unfused projection may differ in their fp32 summation order)
        plain = {o: run(slot - o, None, split_pass) for o in set(off.tolist())}
        q, kc, vc = run(slot, off.cuda(), split_pass)
        for m in range(M):
            o = int(off[m])
            qp, kp, vp = plain[o]
            assert torch.equal(q[m], qp[m]), (m, o, split_pass)
            assert torch.equal(kc[m, :, slot], kp[m, :, slot - o]) and torch.equal(vc[m, :, slot], vp[m, :, slot - o]), (m, o, split_pass)
        keep = torch.ones(Tmax, dtype=torch.bool); keep[slot] = False
        assert kc[:, :, keep].abs().sum() == 0 and vc[:, :, keep].abs().sum() == 0        # nothing outside the common slot
    with pytest.raises(Exception):                       # the prefill form has no row offsets (its caller advances the cache pointers)
        kc = torch.zeros(1, Hk, Tmax, d, dtype=BF, device="cuda")
        ops.gemm(x[:1].expand(8, K).contiguous(), w, rope=(tab, kc, kc.clone(), H, Hk, d, Tmax, 0, None, 8, None), rope_row_off=off[:8].cuda())


@pytest.mark.parametrize("M", [1, 8, 16, 40, 300, 448, 2048])
@pytest.mark.parametrize("with_route", [False, True])
def test_fp32_norm_weights_equal_the_same_values_in_bf16(M, with_route):
    """crab_gemm_desc.norm_w_fp32 (ABI 9): the RMSNorm weight behind o_proj / down_proj held in fp32.  With bf16-representable VALUES the fp32
    storage must give bit-identical rows to the bf16 storage in every regime that owns a fused or stand-alone post-norm - the M <= 16 tail
    (rowfin.hip, with the projection's own adapter riding along), the row-owning split-K reduction (16 < M <= 512, with the next group's router),
    the stand-alone norm behind the large-M kernels - and with genuinely fp32 values it must match fp32 arithmetic; a bf16 residual stream is refused."""
    from crab_amd import ops
    K, N, nl, r = 1024, 2048, 3, 8
    x, w, res = _rand(M, K, seed=1).cuda(), _rand(N, K, seed=2, scale=K ** -0.5).cuda(), _rand(M, N, seed=3)
    g = torch.Generator().manual_seed(9)
    nw16 = (1 + 0.1 * torch.randn(N, generator=g)).to(BF)
    RA = torch.zeros(16, K, dtype=BF); RA[:nl + r] = _rand(nl + r, K, seed=5, scale=K ** -0.5)
    B2 = torch.zeros(N, 32, dtype=BF); B2[:, :nl * r] = _rand(N, nl * r, seed=6, scale=0.2)
    ra_next = _rand(48, N, seed=7, scale=N ** -0.5); ra_next[33:] = 0
    outs = []
    for nw in (nw16.cuda(), nw16.float().cuda()):
        xd = res.float().cuda()
        h = torch.empty(M, N, dtype=BF, device="cuda")
        u = torch.zeros((M, 96), dtype=BF, device="cuda")
        kw = {}
        if with_route and M <= ops.DECODE_MAX_ROWS:
            kw["route"] = (ra_next.cuda(), 3, nl, r, 96, 2.0, u)
        if M <= 16:
            kw["lora_self"] = (RA.cuda(), nl, r, 2.0, B2.cuda())
        ops.gemm(x, w, residual=xd, out=xd, post_norm=(nw, 1e-5, h), **kw)
        outs.append((xd.clone(), h.clone(), u.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b), "fp32-stored norm weights changed the result although their values are bf16-representable"
    nw32 = 1 + 0.1 * torch.randn(N, generator=g)                                   # genuinely fp32 values
    xd = res.float().cuda()
    h = torch.empty(M, N, dtype=BF, device="cuda")
    ops.gemm(x, w, residual=xd, out=xd, post_norm=(nw32.cuda(), 1e-5, h))
    xr = xd.float().cpu()
    _cmp(h, xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * nw32, 4.5e-3, f"post-norm with fp32 norm weights, M={M}")
    _cmp(ops.rmsnorm(xd, nw32.cuda(), 1e-5), xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * nw32, 4.5e-3, f"stand-alone rmsnorm, fp32 weights, M={M}")
    with pytest.raises(Exception):
        xb = res.cuda().clone()
        ops.gemm(x, w, residual=xb, out=xb, post_norm=(nw32.cuda(), 1e-5, h))       # fp32 norm weights need the fp32 residual stream


@pytest.mark.gpu
@pytest.mark.parametrize("M,K,nproj,nl,r", [(300, 11008, 3, 3, 8), (301, 11008, 1, 3, 8), (512, 8200, 2, 3, 8), (257, 12288, 3, 4, 4), (512, 4096, 3, 3, 8), (448, 18944, 1, 3, 8)])
def test_router_two_rows_per_block_equals_one_row_per_block(M, K, nproj, nl, r):
    """lora_route_row2_kernel (256 < M <= 512 and K > 8192: two rows per block share every [R;A] chunk, csrc/skinny.hip) against lora_route_row_kernel forced by
    CRAB_ROUTE_ROWS=1: bit-identical U (same per-row arithmetic and order), odd M and the pad columns included; and against fp32 torch
    (peft_hyper/tuners/lora.py:346-350).  K = 18944 (Qwen2's down projection) is beyond the row kernels' 12288: both calls take the general pair."""
    import os
    from crab_amd import ops
    torch.manual_seed(M + K)
    x = (torch.randn(M, K, device="cuda") * 0.7).to(BF)
    tcols = (nproj * (nl + r) + 15) // 16 * 16
    ra = torch.zeros(tcols, K, device="cuda", dtype=BF)
    ra[: nproj * (nl + r)] = (torch.randn(nproj * (nl + r), K, device="cuda") * 0.02).to(BF)
    ucols = (nproj * nl * r + 7) // 8 * 8 + 8
    outs = []
    for env in (None, "1"):
        if env is None:
            os.environ.pop("CRAB_ROUTE_ROWS", None)
        else:
            os.environ["CRAB_ROUTE_ROWS"] = env
        try:
            u = torch.full((M, ucols), 7.0, device="cuda", dtype=BF)
            ops.hyperlora_route(x, ra, nproj, nl, r, ucols, 2.0, out=u)
            torch.cuda.synchronize()
            outs.append(u.clone())
        finally:
            os.environ.pop("CRAB_ROUTE_ROWS", None)
    assert torch.equal(outs[0], outs[1])
    t = x.float() @ ra.float().t()
    ref = torch.zeros(M, ucols, device="cuda")
    for pj in range(nproj):
        tt = t[:, pj * (nl + r):(pj + 1) * (nl + r)]
        p = torch.softmax(tt[:, :nl], -1)
        ref[:, pj * nl * r:(pj + 1) * nl * r] = (2.0 * p[:, :, None] * tt[:, None, nl:]).reshape(M, nl * r)
    assert (outs[0].float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item() + 1e-6
