"""Differential fuzz of the VQGAN mask tokenizer (crab_amd/vqgan.py) on the reference-recorded tiny configuration (tests/golden/vqgan_tiny.npz
holds its weights by seed): random batch sizes and mask sizes (square, non-square, smaller and larger than the configured resolution, any multiple of the
encoder's stride), against oracle/vqgan_oracle.py: codebook ids EQUAL wherever the oracle's own fp32 top-2 distance gap exceeds 1e-4 of the distance scale
(index work; the precise encoder keeps latents to ~1e-5), the decoded image within 2.5 x the oracle's bf16-storage emulation, a batch equal to its samples
one by one bit for bit.  (The shipped precise encoder, CRAB_VQ_PRECISE=1; the bf16-operand form flips ids at its floor, tests/test_vqgan.py.)   python scripts/fuzz_vqgan.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import vqgan_oracle as VO
from tests.util import load_fixture, weights_from_table, strip
from tests.test_vqgan import _build

_argv = sys.argv[1:] if __name__ == "__main__" else []
NCASE = int(_argv[0]) if len(_argv) > 0 else 12
rng = random.Random(int(_argv[1]) if len(_argv) > 1 else 0)
meta, A = load_fixture("vqgan_tiny")
c = meta["cfg"]
cfg = VO.VQConfig(ch=c["ch"], ch_mult=tuple(c["ch_mult"]), num_res_blocks=c["num_res_blocks"], attn_resolutions=tuple(c["attn_resolutions"]),
                  resolution=c["resolution"], z_channels=c["z_channels"], n_embed=c["n_embed"], embed_dim=c["embed_dim"])
W = weights_from_table(meta)
Wv = strip(W, "mask_encoder.vqgan.")
m = _build(meta, W)
stride = 2 ** (len(c["ch_mult"]) - 1)
bad, worst_lat, worst_dec, n_ids, n_near = [], 0.0, 0.0, 0, 0
for case in range(NCASE):
    B = rng.choice([1, 2, 3, 5])
    H, Wd = stride * rng.choice([1, 2, 4, 8]), stride * rng.choice([1, 2, 3, 4, 8])
    if H == stride and Wd == stride: Wd = 2 * stride          # a 1 x 1 latent map normalises groups of a handful of values: the oracle's own bf16 emulation is 30-60 % off there
    g = torch.Generator().manual_seed(100 + case)
    x = torch.randn(B, 3, H, Wd, generator=g) * rng.choice([0.5, 1.0])
    desc = f"case {case}: B={B} {H}x{Wd}"
    try:
        z, hh, ww = m.vqgan.encode_latents(x.cuda())
        lat = z.cpu().view(B, hh, ww, -1).permute(0, 3, 1, 2)
        shifted = m.encode_mask(x.cuda())
    except Exception as e:      # noqa: BLE001
        bad.append(desc + f" -> {type(e).__name__}: {str(e)[:200]}"); continue
    ref_lat = VO.encode_latents(x, Wv, cfg)
    el = float((lat - ref_lat).abs().max()) / float(ref_lat.abs().max())
    worst_lat = max(worst_lat, el)
    ref_ids = VO.quantize_indices(ref_lat, Wv).reshape(B, -1)
    got_ids = (shifted.cpu() - 32020).reshape(B, -1)
    # the oracle's own margins: distance gap between its best and second-best code
    e = Wv["quantize.embedding.weight"].float()
    zf = ref_lat.permute(0, 2, 3, 1).reshape(-1, e.shape[1])
    d = (zf ** 2).sum(1, keepdim=True) + (e ** 2).sum(1) - 2 * zf @ e.t()
    top2 = d.topk(2, dim=1, largest=False).values
    near = ((top2[:, 1] - top2[:, 0]) < 1e-4 * (1 + d.abs().max())).reshape(B, -1)
    n_ids += got_ids.numel(); n_near += int(near.sum())
    wrong = int(((got_ids != ref_ids) & ~near).sum())
    if wrong: bad.append(desc + f" -> {wrong} of {got_ids.numel()} codebook ids differ from the oracle's beyond its own margin")
    # batch == samples one by one
    for b in range(B):
        one = m.encode_mask(x[b:b + 1].cuda()).cpu().reshape(-1) - 32020
        if not torch.equal(one, got_ids[b]): bad.append(desc + f" -> sample {b} alone gives other ids than inside the batch"); break
    # decode (the reference reshapes the ids to a square, vqgan.py:69-75: square masks only)
    if hh == ww:
        try:
            dec = m.decode_mask(shifted).float().cpu()
        except Exception as e:      # noqa: BLE001
            bad.append(desc + f" -> decode_mask: {type(e).__name__}: {str(e)[:200]}"); continue
        ref_dec = VO.decode_code(got_ids, Wv, cfg)
        with VO.emulate("storage"):
            emu_dec = VO.decode_code(got_ids, Wv, cfg)
        sc = float(ref_dec.abs().max())
        err, emu = float((dec - ref_dec).abs().max()) / sc, float((emu_dec - ref_dec).abs().max()) / sc
        worst_dec = max(worst_dec, err)
        if dec.shape != ref_dec.shape or err > max(2.5 * emu, 1e-2): bad.append(desc + f" -> decoded image {err:.3e} of scale (storage emulation {emu:.3e})")
print(f"{NCASE} cases: latents worst {worst_lat:.2e} of scale, decoded worst {worst_dec:.2e}; {n_ids} ids ({n_near} inside the oracle's own margin); {len(bad)} failures")
for b_ in bad[:30]: print("FAIL", b_)
sys.exit(1 if bad else 0)
