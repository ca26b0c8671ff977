"""r06 probe: the VQGAN tiny fixture through crab_amd.vqgan - latents / decoded error against the reference fixture and the id mismatches, to
see what fp32 GroupNorm parameters + the fp32 quantiser buy (run on the GPU box: python scripts/exp/vqgan_probe.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from tests.test_vqgan import _build, _setup
from tests.util import strip

meta, A, cfg, W, x = _setup()
m = _build(meta, W)
z, hh, ww = m.vqgan.encode_latents(x.cuda())
lat = z.float().cpu().view(2, hh, ww, -1).permute(0, 3, 1, 2)
sc = A["latents"].abs().max()
print("latents rel err", float((lat - A["latents"]).abs().max() / sc), "dtype", z.dtype)
ids = m.encode_mask(x).cpu() - 32020
ref = A["indices"].long()
bad = ids != ref
print("id mismatches", int(bad.sum()), "of", bad.numel(), "margins at mismatches", A["margin"][bad].tolist(), "min margin", float(A["margin"].min()))
# ids from the REFERENCE latents through the fp32 quantiser: must equal the reference's ids
e = strip(W, "mask_encoder.vqgan.")["quantize.embedding.weight"].float()
zr = A["latents"].permute(0, 2, 3, 1).reshape(-1, e.shape[1]).contiguous()
from crab_amd import ops
i2 = ops.vq_nearest_f32(zr.cuda(), e.cuda(), ops.row_sqnorm_f32(e.cuda())).cpu().view(2, -1)
print("fp32 quantiser on the reference's latents: mismatches", int((i2 != ref).sum()))
img = m.decode_mask(ref.cuda() + 32020).cpu()
print("decoded rel err", float((img - A["decoded"]).abs().max() / A["decoded"].abs().max()))
