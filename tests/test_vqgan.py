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


# fraction of the fixture's positions whose id may differ from the reference's (each must also sit at a sub-margin position):
# 2x the fraction measured on MI355X (r03: 1 of 128 positions, recorded in the parity report by this test)
VQ_ID_MISMATCH_CAP = 2 / 128


@pytest.mark.gpu
def test_mask_encoder_matches_reference_fixture_and_oracle():
    meta, A, cfg, W, x = _setup()
    m = _build(meta, W)
    Wv = strip(W, "mask_encoder.vqgan.")
    # latents (encoder + quant_conv) against the reference
    z, hh, ww = m.vqgan.encode_latents(x.cuda())
    lat = z.float().cpu().view(2, hh, ww, -1).permute(0, 3, 1, 2)
    assert _rel(lat, A["latents"]) < 4e-2, _rel(lat, A["latents"])
    # ids: identical wherever the reference's nearest / second-nearest margin exceeds the bf16 distance error
    ids = m.encode_mask(x).cpu() - 32020
    ref = A["indices"].long()
    e = Wv["quantize.embedding.weight"].float()
    zf = z.float().cpu()
    zr = A["latents"].permute(0, 2, 3, 1).reshape(-1, e.shape[1])
    derr = 2 * ((zf - zr) @ e.t()).abs().max(1).values.view(2, -1)         # how far the bf16 latents move any distance
    bad = ids != ref
    from tests.util import record_parity
    record_parity("vqgan_tiny: fraction of codebook ids that differ from the reference (all at sub-margin positions)", bad.float().mean().item(), 1.0,
                  VQ_ID_MISMATCH_CAP, positions=int(bad.numel()), mismatches=int(bad.sum()))
    assert bad.float().mean() <= VQ_ID_MISMATCH_CAP, bad.float().mean()
    assert (A["margin"][bad] <= 2 * derr[bad] + 1e-3).all(), "id mismatch that the latent error cannot explain"
    # decode path from the REFERENCE ids
    img = m.decode_mask(ref.cuda() + 32020).cpu()
    assert img.shape == A["decoded"].shape and _rel(img, A["decoded"]) < 4e-2, _rel(img, A["decoded"])
    # round trip through the public methods, ids clipped like the reference (indices below token_shift -> entry 0)
    img2 = m.decode_mask(m.encode_mask(x))
    assert img2.shape == (2, 3, cfg.resolution, cfg.resolution)
    low = torch.full((1, 16), 5, dtype=torch.long)
    assert torch.equal(m.decode_mask(low.cuda()), m.decode_mask(torch.full((1, 16), 32020, dtype=torch.long).cuda()))


@pytest.mark.gpu
def test_full_size_vqgan_shapes_and_determinism():
    """The taming f16 / 16384 architecture at 256x256: 256 ids per mask, deterministic, decode returns [b,3,256,256]."""
    from crab_amd.vqgan import MaskEncoder
    from crab_amd import synth
    m = MaskEncoder(token_shift=32020)
    sd = {k: synth.synth_tensor("mask_encoder." + k, list(v.shape), 7) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    x = torch.randn(2, 3, 256, 256, generator=torch.Generator().manual_seed(1))
    ids = m.encode_mask(x)
    assert ids.shape == (2, 256) and int(ids.min()) >= 32020 and int(ids.max()) < 32020 + 16384
    assert torch.equal(ids, m.encode_mask(x))
    # batch rows are independent; a different M picks different GEMM tilings (summation order), so near-ties may flip
    assert (ids[1:] == m.encode_mask(x[1:])).float().mean() > 0.97
    img = m.decode_mask(ids)
    assert img.shape == (2, 3, 256, 256) and torch.isfinite(img).all()
