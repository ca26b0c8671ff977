"""Differential fuzz of crab_gemm_bf16 across its kernel regimes (rowfin / skinny M <= 16, 128^2 LDS-DMA, 256^2 ring, split-K, panel kernels for
128 < M <= 512) and epilogues (bias, activations, SwiGLU pairs, bf16 / fp32 residual, fp32 output, K extension, fused post-RMSNorm + next-group
router): random shapes around every dispatch boundary against fp32 torch on the same bf16 operands.  A combination the library rejects must be
rejected with CRAB_E_UNSUPPORTED / CRAB_E_INVALID (counted), never computed wrong.   python scripts/fuzz_gemm.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from crab_amd import ops, _lib

BF = torch.bfloat16
NCASE = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
Ms = [1, 2, 3, 8, 15, 16, 17, 33, 64, 65, 100, 128, 129, 200, 255, 256, 257, 300, 448, 511, 512, 513, 700, 1024, 2050]
Ns = [8, 16, 64, 96, 104, 128, 200, 256, 384, 1000, 1024, 2048, 4096, 4104, 11008]
Ks = [8, 32, 64, 72, 128, 256, 1000, 1024, 4096]
K2s = [0, 0, 8, 32, 64, 96]
ACTS = ["none", "none", "gelu", "quick_gelu", "relu", "silu", "swiglu_pair"]


def ref_act(y, act):
    if act == "gelu": return F.gelu(y)
    if act == "quick_gelu": return y * torch.sigmoid(1.702 * y)
    if act == "relu": return F.relu(y)
    if act == "silu": return F.silu(y)
    if act == "swiglu_pair": return F.silu(y[:, 0::2]) * y[:, 1::2]
    return y


bad, rejected, done, why = [], 0, 0, {}
for case in range(NCASE):
    M, N, K, K2, act = rng.choice(Ms), rng.choice(Ns), rng.choice(Ks), rng.choice(K2s), rng.choice(ACTS)
    if M * N * K > 2050 * 4104 * 4096: continue
    bias = rng.random() < 0.5
    res = rng.choice([None, None, "bf16", "fp32"])
    out32 = (res == "fp32") or rng.random() < 0.2
    norm = M <= 512 and act != "swiglu_pair" and N <= 8192 and N % 8 == 0 and rng.random() < 0.3
    route = norm and rng.random() < 0.5
    g = torch.Generator(device="cuda").manual_seed(case)
    x = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(BF)
    w = (torch.randn(N, K, device="cuda", generator=g) * (1.0 / K ** 0.5)).to(BF)
    x2 = (torch.randn(M, K2, device="cuda", generator=g) * 0.5).to(BF) if K2 else None
    w2 = (torch.randn(N, K2, device="cuda", generator=g) * 0.1).to(BF) if K2 else None
    b = (torch.randn(N, device="cuda", generator=g) * 0.1).to(BF) if bias else None
    No = N // 2 if act == "swiglu_pair" else N
    r = None
    if res:
        r = torch.randn(M, No, device="cuda", generator=g)
        r = r if res == "fp32" else r.to(BF)
    # the output sits inside a larger buffer of sentinels: a guard row above and below, and (half of the cases) 8 / 16 guard columns behind every
    # row - a store outside the M x N block shows up as a changed sentinel
    gap = rng.choice([0, 0, 8, 16])
    obuf = torch.full((M + 2, No + gap), 777.0, device="cuda", dtype=torch.float32 if out32 else BF)
    out = obuf[1:M + 1, :No]
    y = x.float() @ w.float().t()
    if K2: y = y + x2.float() @ w2.float().t()
    if bias: y = y + b.float()
    y = ref_act(y, act)
    if r is not None: y = y + r.float()
    kw = {}
    if norm:
        nw = (1.0 + 0.1 * torch.randn(N, device="cuda", generator=g)).to(BF)
        hbuf = torch.full((M + 2, N + gap), 777.0, device="cuda", dtype=BF)
        h = hbuf[1:M + 1, :N]
        kw["post_norm"] = (nw, 1e-5, h)
        if route:
            ra = (torch.randn(48, N, device="cuda", generator=g) * 0.05).to(BF)
            ubuf = torch.full((M + 2, 96 + 8), 777.0, device="cuda", dtype=BF)
            u = ubuf[1:M + 1, :96]
            kw["route"] = (ra, 3, 3, 8, 96, 2.0, u)
    desc = f"case {case}: M={M} N={N} K={K}+{K2} act={act} bias={bias} res={res} out={'fp32' if out32 else 'bf16'} norm={norm} route={route}"
    try:
        ops.gemm(x, w, bias=b, act=act, residual=r, x2=x2, w2=w2, out=out, **kw)
        torch.cuda.synchronize()
    except _lib.CrabHipError as e:
        msg = str(e)
        if "error -1:" in msg or "error -3:" in msg:              # CRAB_E_INVALID / CRAB_E_UNSUPPORTED: a stated limit, not a wrong answer
            rejected += 1
            why[msg.split(":", 2)[-1].strip()[:90]] = why.get(msg.split(":", 2)[-1].strip()[:90], 0) + 1
            continue
        bad.append(desc + " -> " + msg[:200]); continue
    done += 1
    guards = [("out", obuf, No)] + ([("post-norm out", hbuf, N)] if norm else []) + ([("router out", ubuf, 96)] if route else [])
    for gname, gb, cols in guards:
        if not (bool((gb[0] == 777.0).all()) and bool((gb[-1] == 777.0).all()) and bool((gb[:, cols:] == 777.0).all())):
            bad.append(desc + f" gap={gap} -> {gname}: a store landed outside the M x N block")
    scale = float(y.abs().max()) + 1e-6
    err = float((out.float() - y).abs().max()) / scale
    tol = 6e-3 if out32 else 1.2e-2
    if not (err < tol) or not torch.isfinite(out.float()).all():
        bad.append(desc + f" -> rel err {err:.3e} (tol {tol})")
    if norm:
        stored = out.float()
        hh = stored * torch.rsqrt(stored.pow(2).mean(-1, keepdim=True) + 1e-5) * nw.float()
        e2 = float((h.float() - hh).abs().max()) / (float(hh.abs().max()) + 1e-6)
        if not (e2 < 1.2e-2): bad.append(desc + f" -> post-norm rel err {e2:.3e}")
        if route:
            t = h.float() @ ra.float().t()
            uu = torch.zeros(M, 96, device="cuda")
            for p in range(3):
                tt = t[:, p * 11:(p + 1) * 11]
                pr = torch.softmax(tt[:, :3], -1)
                uu[:, p * 24:(p + 1) * 24] = (2.0 * pr[:, :, None] * tt[:, None, 3:]).reshape(M, 24)
            e3 = float((u.float() - uu).abs().max()) / (float(uu.abs().max()) + 1e-6)
            if not (e3 < 2e-2): bad.append(desc + f" -> router rel err {e3:.3e}")
print(f"{done} cases computed, {rejected} rejected by the library, {len(bad)} failures")
for k_, v_ in sorted(why.items(), key=lambda kv: -kv[1]): print(f"  rejected x{v_}: {k_}")
for b_ in bad[:40]: print("FAIL", b_)
sys.exit(1 if bad else 0)
