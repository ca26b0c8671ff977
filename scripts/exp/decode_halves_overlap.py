"""Experiment: a decode step as TWO half batches that ping-pong between a stream that runs the (HBM-bound) decode attention and a
stream that runs the (L2 / LDS-DMA-bound) projections, on disjoint CU sets (hipExtStreamCreateWithCUMask) - against the shipped
order (one batch, attention then projections, whole GPU).  Per "layer": attention over ctx tokens + q|k|v, o, gate|up, down.
usage: decode_halves_overlap.py [clips] [n_attn_cus ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crab_amd import ops
hip = C.CDLL("libamdhip64.so")
BF = torch.bfloat16


def masked_stream(bits):
    words = [0] * 8
    for b in bits: words[b // 32] |= 1 << (b % 32)
    arr = (C.c_uint32 * 8)(*words)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


B = int(sys.argv[1]) if len(sys.argv) > 1 else 448
NAS = [int(a) for a in sys.argv[2:]] or [128, 160, 192]
ctx, H, d, Tmax, D, I = 830, 32, 128, 960, 4096, 11008
kc = (torch.randn(B, H, Tmax, d, device="cuda") * 0.5).to(BF); vc = (torch.randn(B, H, Tmax, d, device="cuda") * 0.5).to(BF)
q = torch.randn(B, 3 * H * d, device="cuda").to(BF); att = torch.empty(B, D, device="cuda", dtype=BF)
NL = 4                                                       # weight sets rotated so that every projection streams from HBM
W = [dict(qkv=(torch.randn(3 * D, D, device="cuda") * 0.02).to(BF), qkv2=(torch.randn(3 * D, 96, device="cuda") * 0.02).to(BF),
          o=(torch.randn(D, D, device="cuda") * 0.02).to(BF), o2=(torch.randn(D, 32, device="cuda") * 0.02).to(BF),
          gu=(torch.randn(2 * I, D, device="cuda") * 0.02).to(BF), gu2=(torch.randn(2 * I, 64, device="cuda") * 0.02).to(BF),
          down=(torch.randn(D, I, device="cuda") * 0.02).to(BF), down2=(torch.randn(D, 32, device="cuda") * 0.02).to(BF)) for _ in range(NL)]


class Half:
    def __init__(self, b0, n):
        self.b0, self.n = b0, n
        self.h = torch.randn(n, D, device="cuda").to(BF); self.u = torch.randn(n, 96, device="cuda").to(BF)
        self.x = torch.randn(n, D, device="cuda"); self.act = torch.empty(n, I, device="cuda", dtype=BF)
        self.q = q[b0:b0 + n]; self.att = att[b0:b0 + n]; self.kc = kc[b0:b0 + n]; self.vc = vc[b0:b0 + n]

    def attn(self):
        ops.attn_decode(self.q, self.kc, self.vc, self.att, self.n, H, H, d, Tmax, ctx, d ** -0.5)

    def proj(self, l):
        w = W[l % NL]
        ops.gemm(self.att, w["o"], x2=self.u[:, :32], w2=w["o2"], residual=self.x, out=self.x)
        ops.gemm(self.h, w["gu"], x2=self.u[:, :64], w2=w["gu2"], act="swiglu_pair", out=self.act)
        ops.gemm(self.act, w["down"], x2=self.u[:, :32], w2=w["down2"], residual=self.x, out=self.x)
        ops.gemm(self.h, w["qkv"], x2=self.u, w2=w["qkv2"], out=self.q)


def timed(fn):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)


full = Half(0, B); ha, hb = Half(0, B // 2), Half(B // 2, B - B // 2)
NLAY = 16
for hh in (full, ha, hb):
    hh.attn(); hh.proj(0)
def serial_full():
    for l in range(NLAY): full.attn(); full.proj(l)
def serial_halves():
    for l in range(NLAY):
        ha.attn(); ha.proj(l); hb.attn(); hb.proj(l)
serial_full(); serial_halves()
t_full = timed(serial_full) / NLAY * 1e3
t_attn = timed(lambda: [full.attn() for _ in range(NLAY)]) / NLAY * 1e3
t_proj = timed(lambda: [full.proj(l) for l in range(NLAY)]) / NLAY * 1e3
t_halves = timed(serial_halves) / NLAY * 1e3
print(f"B={B}: one batch, whole GPU: {t_full:.0f} us per layer (attention {t_attn:.0f} + projections {t_proj:.0f}); two halves back to back {t_halves:.0f} us", flush=True)


def pingpong(sa, sb):
    """half A: attn on sa -> proj on sb -> attn on sa ...; half B the same, half a layer behind: sa and sb are both always busy"""
    cur = torch.cuda.current_stream()
    sa.wait_stream(cur); sb.wait_stream(cur)
    ev = {}
    def attn(hh, key, after):
        with torch.cuda.stream(sa):
            if after is not None: sa.wait_event(after)
            hh.attn(); e = torch.cuda.Event(); e.record(sa); return e
    def proj(hh, l, after):
        with torch.cuda.stream(sb):
            sb.wait_event(after)
            hh.proj(l); e = torch.cuda.Event(); e.record(sb); return e
    pa = pb = None
    for l in range(NLAY):
        ea = attn(ha, "a", pa)
        eb = attn(hb, "b", pb)
        pa = proj(ha, l, ea)
        pb = proj(hb, l, eb)
    cur.wait_stream(sa); cur.wait_stream(sb)


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
pingpong(s1, s2)
t = timed(lambda: pingpong(s1, s2)) / NLAY * 1e3
print(f"two plain streams (no CU mask): {t:.0f} us per layer ({t / t_full:.3f} x)", flush=True)
for NA in NAS:
    for layout in ("low", "interleaved"):
        if layout == "low": abits = list(range(NA))
        else: abits = [i for i in range(256) if (i % 8) < (NA * 8 // 256)]
        aset = set(abits)
        gbits = [i for i in range(256) if i not in aset]
        sa, sb = masked_stream(abits), masked_stream(gbits)
        pingpong(sa, sb)
        t = timed(lambda: pingpong(sa, sb)) / NLAY * 1e3
        with torch.cuda.stream(sa):
            ta = timed(lambda: [ha.attn() for _ in range(NLAY)]) / NLAY * 1e3
        with torch.cuda.stream(sb):
            tp = timed(lambda: [ha.proj(l) for l in range(NLAY)]) / NLAY * 1e3
        print(f"attention on {len(abits)} CUs ({layout}), projections on {len(gbits)}: ping-pong {t:.0f} us per layer ({t / t_full:.3f} x of the shipped order); "
              f"alone: half attention {ta:.0f} us, half projections {tp:.0f} us", flush=True)
