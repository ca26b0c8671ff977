"""VQGAN mask tokenizer: oracle vs the reference-recorded fixture on CPU, HIP kernels and the MaskEncoder
mirror on the GPU."""
import pytest
import torch
import torch.nn.functional as F

from oracle import vqgan_oracle as VO
from tests.util import load_fixture, weights_from_table, strip

BF = torch.bfloat16


def _setup():
    meta, A = load_fixture("vqgan_tiny")
    c = meta["cfg"]
    cfg = VO.VQConfig(ch=c["ch"], ch_mult=tuple(c["ch_mult"]), num_res_blocks=c["num_res_blocks"], attn_resolutions=tuple(c["attn_resolutions"]),
                      resolution=c["resolution"], z_channels=c["z_channels"], n_embed=c["n_embed"], embed_dim=c["embed_dim"])
    W = weights_from_table(meta)
    g = torch.Generator().manual_seed(meta["xseed"])
    x = torch.randn(2, 3, c["resolution"], c["resolution"], generator=g)
    return meta, A, cfg, W, x


def test_vqgan_oracle_matches_reference_fixture():
    meta, A, cfg, W, x = _setup()
    Wv = strip(W, "mask_encoder.vqgan.")
    lat = VO.encode_latents(x, Wv, cfg)
    assert (lat - A["latents"]).abs().max() < 2e-5
    idx = VO.get_codebook_indices(x, Wv, cfg)
    assert torch.equal(idx, A["indices"].long())
    dec = VO.decode_code(idx, Wv, cfg)
    assert (dec - A["decoded"]).abs().max() < 2e-5
    shifted = VO.encode_mask(x, Wv, cfg, 32020)
    assert torch.equal(VO.decode_mask(shifted, Wv, cfg, 32020), dec)


def _rel(a, b, what=""):
    from tests.util import rel_err
    return rel_err(a, b, what)


@pytest.mark.gpu
def test_vq_kernels_vs_torch():
    from crab_amd import ops
    g = torch.Generator().manual_seed(4)
    B, h, w, C = 2, 12, 10, 64
    x = torch.randn(B * h * w, C, generator=g).to(BF)
    xi = x.float().view(B, h, w, C).permute(0, 3, 1, 2)
    # strided window (Downsample: pad right/bottom by one, stride 2) and the generic pad-1 stride-1 form
    oh, ow = (h + 1 - 3) // 2 + 1, (w + 1 - 3) // 2 + 1
    cols = ops.im2col3x3_strided(x.cuda(), B, h, w, 2, 0, 0, oh, ow).cpu().float()
    ref = F.unfold(F.pad(xi, (0, 1, 0, 1)), 3, stride=2).view(B, C, 9, oh * ow).permute(0, 3, 2, 1).reshape(B * oh * ow, 9 * C)
    assert torch.equal(cols, ref)
    cols1 = ops.im2col3x3_strided(x.cuda(), B, h, w, 1, 1, 1, h, w)
    assert torch.equal(cols1, ops.im2col3x3(x.cuda(), B, h, w))
    # GroupNorm(32) + swish, channels per group 2 and 16, many pixel chunks
    for (HW, Cg) in ((h * w, 64), (70 * 70, 512), (7, 128)):
        xx = (torch.randn(B * HW, Cg, generator=g) * 2 + 0.5).to(BF)
        wt, bs = (1 + 0.1 * torch.randn(Cg, generator=g)).to(BF), (0.1 * torch.randn(Cg, generator=g)).to(BF)
        r = F.group_norm(xx.float().view(B, HW, Cg).permute(0, 2, 1), 32, wt.float(), bs.float(), 1e-6).permute(0, 2, 1).reshape(B * HW, Cg)
        for sw in (False, True):
            y = ops.groupnorm(xx.cuda(), B, HW, 32, wt.cuda(), bs.cuda(), 1e-6, sw)
            assert _rel(y, r * torch.sigmoid(r) if sw else r) < 1e-2
    # nearest 2x
    up = ops.upsample_nearest2x(x.cuda(), B, h, w).cpu().float().view(B, 2 * h, 2 * w, C).permute(0, 3, 1, 2)
    assert torch.equal(up, F.interpolate(xi, scale_factor=2.0, mode="nearest"))
    # row softmax
    s = torch.randn(300, 257, generator=g) * 5
    assert _rel(ops.softmax_rows(s.cuda(), 0.37), torch.softmax(s * 0.37, 1)) < 1e-2
    # codebook norms + argmin: exact, first minimum wins on ties
    e = torch.randn(1000, 64, generator=g).to(BF)
    e[777] = e[5]                                                  # duplicate entry: 5 must win over 777
    z = torch.cat([e[[5, 123, 999]], torch.randn(61, 64, generator=g).to(BF)])
    e2 = ops.row_sqnorm(e.cuda())
    assert _rel(e2, (e.float() ** 2).sum(1)) < 1e-6
    dots = ops.gemm(z.cuda(), e.cuda(), out_fp32=True)
    idx = ops.vq_argmin(dots, e2, offset=100).cpu()
    d = e2.cpu()[None] - 2 * dots.cpu()
    assert torch.equal(idx, torch.argmin(d, 1) + 100) and idx[:3].tolist() == [105, 223, 1099]


def _build(meta, W, device="cuda"):
    from crab_amd.vqgan import MaskEncoder
    c = meta["cfg"]
    dd = dict(double_z=False, z_channels=c["z_channels"], resolution=c["resolution"], in_channels=3, out_ch=3, ch=c["ch"],
              ch_mult=tuple(c["ch_mult"]), num_res_blocks=c["num_res_blocks"], attn_resolutions=tuple(c["attn_resolutions"]), dropout=0.0)
    m = MaskEncoder(token_shift=32020, device=device, ddconfig=dd, n_embed=c["n_embed"], embed_dim=c["embed_dim"])
    r = m.load_state_dict(strip(W, "mask_encoder."), strict=True)
    return m


def _id_audit(ids, ref_ids, lat, ref_lat, e, margin):
    """ids vs the reference's: (mismatches, all of them explained by the latents' error).  A flip is explained when the reference's own nearest /
    second-nearest margin at that position is below twice the largest distance change the latent error can cause there."""
    zf = lat.permute(0, 2, 3, 1).reshape(-1, e.shape[1])
    zr = ref_lat.permute(0, 2, 3, 1).reshape(-1, e.shape[1])
    derr = 2 * ((zf - zr) @ e.t()).abs().max(1).values.view(ids.shape[0], -1)
    bad = ids != ref_ids
    return int(bad.sum()), bool((margin[bad] <= 2 * derr[bad] + 1e-6).all())


@pytest.mark.gpu
def test_precise_kernels_vs_torch():
    """The pieces of the precise encoder (csrc/vq_ops.hip, r06): the split-operand product against an fp64 product (2^-16-class error where plain
    bf16 operands give 2^-8), fp32 GroupNorm / softmax / bias kernels, and the fp32 quantiser against torch's own fp32 expression."""
    from crab_amd import ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(300, 200, generator=g)
    w = torch.randn(96, 200, generator=g) * 0.1
    ref = (x.double() @ w.double().t()).float()
    y3 = ops.gemm(ops.split3(x.cuda(), 0), ops.split3(w.cuda(), 1), out_fp32=True).cpu()
    y1 = ops.gemm(x.to(BF).cuda(), w.to(BF).cuda(), out_fp32=True).cpu()
    sc = ref.abs().max()
    e3, e1 = float((y3 - ref).abs().max() / sc), float((y1 - ref).abs().max() / sc)
    from tests.util import record_parity
    record_parity("split-operand GEMM (hi.hi + lo.hi + hi.lo over 3K) vs fp64 product; plain bf16 operands alongside", e3 * float(sc), float(sc), 3e-5, bf16_operands_rel=e1)
    assert e3 < 3e-5 and e1 > 30 * e3, (e3, e1)
    s0 = ops.split3(x.cuda(), 0).cpu().float()
    assert torch.equal(s0[:, :200], x.to(BF).float()) and torch.equal(s0[:, 400:], x.to(BF).float())
    assert float((s0[:, :200] + s0[:, 200:400] - x).abs().max()) < 2 ** -15 * float(x.abs().max())
    assert torch.equal(ops.split3(x.cuda(), 1).cpu().float()[:, 200:400], x.to(BF).float())
    # fp32 GroupNorm (+ swish), fp32 row softmax, fp32 bias
    B, HW, Cg = 2, 70, 64
    xx = torch.randn(B * HW, Cg, generator=g) * 2 + 0.5
    wt, bs = 1 + 0.1 * torch.randn(Cg, generator=g), 0.1 * torch.randn(Cg, generator=g)
    r = F.group_norm(xx.view(B, HW, Cg).permute(0, 2, 1), 32, wt, bs, 1e-6).permute(0, 2, 1).reshape(B * HW, Cg)
    for sw in (False, True):
        y = ops.groupnorm_f32(xx.cuda(), B, HW, 32, wt.cuda(), bs.cuda(), 1e-6, sw).cpu()
        assert (y - (r * torch.sigmoid(r) if sw else r)).abs().max() < 2e-5
    sc_ = torch.randn(33, 257, generator=g) * 5
    assert (ops.softmax_rows_f32(sc_.cuda(), 0.37).cpu() - torch.softmax(sc_ * 0.37, 1)).abs().max() < 1e-6
    yb = xx.clone().cuda()
    assert torch.equal(ops.add_bias_f32(yb, bs.cuda()).cpu(), xx + bs)
    # the quantiser in fp32: equal to torch's fp32 expression (quantize.py:286-290), first minimum on ties, N not a multiple of the tile
    e = torch.randn(1000, 64, generator=g)
    e[777] = e[5]
    z = torch.cat([e[[5, 123, 999]], torch.randn(200, 64, generator=g)])
    d = (z ** 2).sum(1, keepdim=True) + (e ** 2).sum(1) - 2 * z @ e.t()
    idx = ops.vq_nearest_f32(z.cuda(), e.cuda(), ops.row_sqnorm_f32(e.cuda()), offset=100).cpu()
    want = torch.argmin(d, 1) + 100
    top2 = d.topk(2, dim=1, largest=False).values
    tie = (top2[:, 1] - top2[:, 0]) < 1e-4                    # fp32 summation order may decide a near-tie differently; exact ties go to the first index
    assert torch.equal(idx[~tie], want[~tie]) and idx[:3].tolist() == [105, 223, 1099]


@pytest.mark.gpu
def test_mask_encoder_matches_reference_fixture_and_oracle():
    """MaskEncoder.encode_mask / decode_mask against the fixture recorded from the reference's VQModel (models/taming_transformer/vqgan.py).
    r06: codebook ids are INDEX work - the encoder in front of the quantiser runs in its precise form (fp32 activations, split-bf16 MFMA
    operands) and the quantiser in fp32, so on this fixture EVERY id equals the reference's (min margin 4.6e-3 against latents accurate to
    ~1e-5).  The bf16-operand form (CRAB_VQ_PRECISE=0) is audited next to it against bounds COMPUTED here: the oracle's bf16-operand floor
    (what no bf16-MFMA encoder can beat: it flips ids on this fixture itself) and its storage emulation."""
    import crab_amd.vqgan as V
    from tests.util import record_parity
    meta, A, cfg, W, x = _setup()
    Wv = strip(W, "mask_encoder.vqgan.")
    e = Wv["quantize.embedding.weight"].float()
    ref = A["indices"].long()
    sc = A["latents"].abs().max().item()
    # ---- floors from the oracle (CPU, tiny configuration)
    emu = {}
    for mode in ("floor", "storage"):
        with VO.emulate(mode):
            lat_o = VO.encode_latents(x, Wv, cfg)
            dec_o = VO.decode_code(ref, Wv, cfg)
        emu[mode] = ((lat_o - A["latents"]).abs().max().item(), (dec_o - A["decoded"]).abs().max().item(),
                     _id_audit(VO.quantize_indices(lat_o, Wv).reshape(2, -1), ref, lat_o, A["latents"], e, A["margin"])[0])
    assert emu["floor"][2] >= 1, "the fixture no longer separates the bf16 operand floor from exact ids"
    # ---- the shipped (precise) form
    assert V.VQ_PRECISE
    m = _build(meta, W)
    z, hh, ww = m.vqgan.encode_latents(x.cuda())
    assert z.dtype == torch.float32
    lat = z.cpu().view(2, hh, ww, -1).permute(0, 3, 1, 2)
    lerr = (lat - A["latents"]).abs().max().item()
    ids = m.encode_mask(x).cpu() - 32020
    nbad, explained = _id_audit(ids, ref, lat, A["latents"], e, A["margin"])
    record_parity("vqgan_tiny latents, precise encoder (split-bf16 operands) vs the reference fixture", lerr, sc, 1e-4, bf16_operand_floor_abs=emu["floor"][0],
                  id_mismatches=nbad, positions=int(ref.numel()), operand_floor_id_mismatches=emu["floor"][2], min_ref_margin=float(A["margin"].min()))
    assert lerr < 1e-4 * sc, (lerr, sc)
    assert nbad == 0 and torch.equal(ids, ref), "codebook ids differ from the reference's"
    # ---- decode path from the REFERENCE ids (bf16 operands: an image, not index work): bounded by the computed floor / emulation
    img = m.decode_mask(ref.cuda() + 32020).cpu()
    derr = (img - A["decoded"]).abs().max().item()
    dsc = A["decoded"].abs().max().item()
    record_parity("vqgan_tiny decoded image vs the reference fixture", derr, dsc, 1.5 * max(emu["floor"][1], emu["storage"][1]) / dsc,
                  bf16_operand_floor_abs=emu["floor"][1], bf16_storage_emulation_abs=emu["storage"][1])
    assert img.shape == A["decoded"].shape and derr <= 1.5 * max(emu["floor"][1], emu["storage"][1]), (derr, emu)
    # round trip through the public methods, ids clipped like the reference (indices below token_shift -> entry 0)
    img2 = m.decode_mask(m.encode_mask(x))
    assert img2.shape == (2, 3, cfg.resolution, cfg.resolution)
    low = torch.full((1, 16), 5, dtype=torch.long)
    assert torch.equal(m.decode_mask(low.cuda()), m.decode_mask(torch.full((1, 16), 32020, dtype=torch.long).cuda()))
    # ---- A/B: the bf16-operand encoder (r01-r05): at the floor, ids flip only where the latent error explains it
    V.VQ_PRECISE = False
    try:
        mb = _build(meta, W)
    finally:
        V.VQ_PRECISE = True
    zb, _, _ = mb.vqgan.encode_latents(x.cuda())
    latb = zb.float().cpu().view(2, hh, ww, -1).permute(0, 3, 1, 2)
    lerr_b = (latb - A["latents"]).abs().max().item()
    idb = mb.encode_mask(x).cpu() - 32020
    nb, expl = _id_audit(idb, ref, latb, A["latents"], e, A["margin"])
    record_parity("vqgan_tiny latents, bf16-operand encoder (CRAB_VQ_PRECISE=0) vs the reference fixture", lerr_b, sc, 1.5 * max(emu["floor"][0], emu["storage"][0]) / sc,
                  bf16_operand_floor_abs=emu["floor"][0], bf16_storage_emulation_abs=emu["storage"][0], id_mismatches=nb, operand_floor_id_mismatches=emu["floor"][2])
    assert lerr_b <= 1.5 * max(emu["floor"][0], emu["storage"][0]) and expl and nb <= 2 * max(emu["floor"][2], emu["storage"][2], 1), (lerr_b, nb, emu)


@pytest.mark.gpu
def test_full_size_vqgan_vs_oracle():
    """The taming f16 / 16384 architecture at 256 x 256 (multimodal_encoder.py:546-601), one mask, against oracle/vqgan_oracle.py on the same seeded
    weights (fp32, executed with torch on the GPU box): fp32 latents within 2e-4 of their scale, the 256 codebook ids EQUAL wherever the oracle's
    own nearest / second-nearest margin exceeds the latent error (with 16384 random entries the margins are tiny: the audit says which positions
    could not be decided), the decoded image inside the computed operand-floor bound; deterministic, batch rows independent."""
    from crab_amd.vqgan import MaskEncoder
    from crab_amd import synth
    from tests.util import record_parity
    m = MaskEncoder(token_shift=32020)
    sd = {k: synth.synth_tensor("mask_encoder." + k, list(v.shape), 7) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(1))
    ids = m.encode_mask(x)
    assert ids.shape == (2, 256) and int(ids.min()) >= 32020 and int(ids.max()) < 32020 + 16384
    assert torch.equal(ids, m.encode_mask(x))
    dev = torch.device("cuda")
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    Wv = {k[len("vqgan."):]: v.to(dev) for k, v in sd.items()}
    cfg = VO.VQConfig()
    x1 = x[:1].to(dev)
    lat_o = VO.encode_latents(x1, Wv, cfg)
    z, hh, ww = m.vqgan.encode_latents(x1)
    lat = z.view(1, hh, ww, -1).permute(0, 3, 1, 2)
    sc = lat_o.abs().max().item()
    lerr = (lat - lat_o).abs().max().item()
    e = Wv["quantize.embedding.weight"].float()
    zf = lat_o.permute(0, 2, 3, 1).reshape(-1, e.shape[1])
    d = (zf ** 2).sum(1, keepdim=True) + (e ** 2).sum(1) - 2 * zf @ e.t()
    top2 = d.topk(2, dim=1, largest=False).values
    margin = (top2[:, 1] - top2[:, 0]).view(1, -1).cpu()
    ref_ids = torch.argmin(d, 1).view(1, -1).cpu()
    got = (ids[:1].cpu() - 32020)
    nbad, explained = _id_audit(got, ref_ids, lat.cpu(), lat_o.cpu(), e.cpu(), margin)
    record_parity("full-size VQGAN (f16 / 16384) latents, precise encoder vs fp32 oracle", lerr, sc, 2e-4, id_mismatches=nbad, positions=256,
                  min_ref_margin=float(margin.min()), median_ref_margin=float(margin.median()))
    assert lerr < 2e-4 * sc, (lerr, sc)
    assert explained and nbad <= 2, (nbad, explained)
    # decode (bf16 operands) against the oracle, bound from its operand floor
    dec_o = VO.decode_code(ref_ids.to(dev), Wv, cfg)
    with VO.emulate("floor"):
        dec_f = VO.decode_code(ref_ids.to(dev), Wv, cfg)
    with VO.emulate("storage"):
        dec_s = VO.decode_code(ref_ids.to(dev), Wv, cfg)
    img = m.decode_mask(ref_ids.cuda() + 32020)
    assert img.shape == (1, 3, 256, 256) and torch.isfinite(img).all()
    derr, fl, st = (img - dec_o).abs().max().item(), (dec_f - dec_o).abs().max().item(), (dec_s - dec_o).abs().max().item()
    record_parity("full-size VQGAN decoded image vs fp32 oracle", derr, dec_o.abs().max().item(), 1.5 * max(fl, st) / dec_o.abs().max().item(),
                  bf16_operand_floor_abs=fl, bf16_storage_emulation_abs=st)
    assert derr <= 1.5 * max(fl, st), (derr, fl, st)
    # batch rows are independent
    assert (ids[1:] == m.encode_mask(x[1:])).float().mean() > 0.97
