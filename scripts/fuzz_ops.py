"""Differential fuzz of the row kernels around the GEMMs (RMSNorm / LayerNorm on bf16 and fp32 rows, embedding to bf16 / fp32 with skipped and clamped
ids, row casts and copies, the stand-alone hyper-LoRA router, the SwiGLU pass, arg-max with a suppressed id; r06: the fp32 quantiser, the split-operand GEMM
and the fp32 GroupNorm of the precise VQGAN encoder) against torch, every output inside
sentinel guard rows / columns.   python scripts/fuzz_ops.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from crab_amd import ops, _lib

BF = torch.bfloat16
NCASE = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad, rejected, done, why = [], 0, 0, {}
S = 777.0


def guarded(M, N, dtype, gap):
    buf = torch.full((M + 2, N + gap), S, device="cuda", dtype=dtype)
    return buf, buf[1:M + 1, :N]


def intact(buf, N):
    return bool((buf[0] == S).all()) and bool((buf[-1] == S).all()) and bool((buf[:, N:] == S).all())


for case in range(NCASE):
    g = torch.Generator(device="cuda").manual_seed(case)
    kind = rng.choice(["rmsnorm", "layernorm", "embedding", "cast", "copy", "route", "swiglu", "argmax", "vq_nearest", "split_gemm", "groupnorm_f32"])
    M = rng.choice([1, 2, 3, 5, 16, 17, 63, 64, 65, 256, 257, 700, 3001])
    gap = rng.choice([0, 8, 16])
    desc, err, tol, ok_guard = f"case {case}: {kind} M={M}", 0.0, 0.0, True
    try:
        if kind in ("rmsnorm", "layernorm"):
            D = rng.choice([8, 64, 128, 200, 768, 1024, 2048, 4096, 8192])
            f32 = rng.random() < 0.5
            xb = torch.randn(M, D + gap, device="cuda", generator=g) * rng.choice([0.1, 1.0, 30.0]) + rng.choice([0.0, 0.5])
            x = (xb if f32 else xb.to(BF))[:, :D]
            w = (1 + 0.2 * torch.randn(D, device="cuda", generator=g)).to(BF)
            b = (0.1 * torch.randn(D, device="cuda", generator=g)).to(BF) if rng.random() < 0.7 else None
            eps = rng.choice([1e-5, 1e-6, 1e-12])
            buf, out = guarded(M, D, BF, gap)
            desc += f" D={D} in={'fp32' if f32 else 'bf16'} gap={gap} eps={eps}"
            xf = x.float()
            if kind == "rmsnorm":
                ops.rmsnorm(x, w, eps, out=out)
                ref = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * w.float()
            else:
                ops.layernorm(x, w, b, eps, out=out)
                ref = F.layer_norm(xf, (D,), w.float(), b.float() if b is not None else None, eps)
            err, tol = float((out.float() - ref).abs().max()) / (float(ref.abs().max()) + 1e-9), 1.2e-2
            ok_guard = intact(buf, D)
        elif kind == "embedding":
            V, D = rng.choice([10, 320, 32017]), rng.choice([8, 128, 4096])
            table = torch.randn(V, D, device="cuda", generator=g).to(BF)
            ids = torch.randint(-2, V + 3, (M,), device="cuda", generator=g)
            f32 = rng.random() < 0.5
            buf, out = guarded(M, D, torch.float32 if f32 else BF, gap)
            out.fill_(5.0)
            ops.embedding(ids, table, out=out)
            ref = table[ids.clamp(0, V - 1)].float()
            ref[ids < 0] = 5.0                                          # a negative id leaves its row to the modality splice
            err, tol = float((out.float() - ref).abs().max()), 0.0
            desc += f" V={V} D={D} out={'fp32' if f32 else 'bf16'}"
            ok_guard = intact(buf, D)
        elif kind in ("cast", "copy"):
            D = rng.choice([8, 16, 128, 1000, 4096])
            sd, dd = (rng.choice([(BF, torch.float32), (torch.float32, BF)]) if kind == "cast" else (BF, BF))
            src = (torch.randn(M, D + gap, device="cuda", generator=g)).to(sd)[:, :D]
            buf, dst = guarded(M, D, dd, rng.choice([0, 8]))
            (ops.cast_rows if kind == "cast" else ops.copy_rows)(src, dst, M, D)
            err, tol = float((dst.float() - src.to(dd).float()).abs().max()), 0.0
            desc += f" D={D} {sd}->{dd}"
            ok_guard = intact(buf, D)
        elif kind == "route":
            K = rng.choice([64, 128, 1024, 4096, 11008])
            nproj, (nl, r) = rng.choice([1, 2, 3]), rng.choice([(3, 8), (3, 8), (2, 4), (8, 4), (3, 16)])
            rows = nproj * (nl + r)
            ucols = (nproj * nl * r + 7) // 8 * 8 + rng.choice([0, 8])
            x = (torch.randn(M, K, device="cuda", generator=g) * 0.5).to(BF)
            ra = (torch.randn((rows + 15) // 16 * 16, K, device="cuda", generator=g) * K ** -0.5).to(BF)
            buf, u = guarded(M, ucols, BF, 8)
            ops.hyperlora_route(x, ra, nproj, nl, r, ucols, 2.0, out=u)
            t = x.float() @ ra.float().t()
            ref = torch.zeros(M, ucols, device="cuda")
            for p in range(nproj):
                tt = t[:, p * (nl + r):(p + 1) * (nl + r)]
                pr = torch.softmax(tt[:, :nl], -1)
                ref[:, p * nl * r:(p + 1) * nl * r] = (2.0 * pr[:, :, None] * tt[:, None, nl:]).reshape(M, nl * r)
            err, tol = float((u.float() - ref).abs().max()) / (float(ref.abs().max()) + 1e-9), 1.5e-2
            desc += f" K={K} nproj={nproj} nl={nl} r={r} ucols={ucols}"
            ok_guard = intact(buf, ucols)
        elif kind == "swiglu":
            I = rng.choice([4, 64, 1376, 11008])
            gu = torch.randn(M, 2 * I + gap, device="cuda", generator=g).to(BF)[:, :2 * I]
            buf, out = guarded(M, I, BF, gap)
            ops.swiglu(gu, out=out)
            ref = F.silu(gu[:, :I].float()) * gu[:, I:].float()
            err, tol = float((out.float() - ref).abs().max()) / (float(ref.abs().max()) + 1e-9), 1.0e-2
            desc += f" I={I}"
            ok_guard = intact(buf, I)
        elif kind == "vq_nearest":
            # r06: the fp32 quantiser (crab_vq_nearest_f32) against torch's fp32 expression of quantize.py:286-290; ids equal wherever the top-2 gap is above fp32 noise
            D, N = rng.choice([4, 32, 64, 256]), rng.choice([1, 63, 64, 65, 1000, 16384])
            Mq = min(M, 700)
            e = torch.randn(N, D, device="cuda", generator=g)
            z = torch.randn(Mq, D, device="cuda", generator=g)
            if N > 8:
                e[N // 2] = e[1]                                        # a duplicate entry: the first index must win
                z[0] = e[1]
            idx = ops.vq_nearest_f32(z, e, ops.row_sqnorm_f32(e), offset=7)
            d = (z ** 2).sum(1, keepdim=True) + (e ** 2).sum(1) - 2 * z @ e.t()
            want = torch.argmin(d, 1) + 7
            top2 = d.topk(min(2, N), dim=1, largest=False).values
            near = (top2[:, -1] - top2[:, 0]) < 1e-4 * (1 + d.abs().max()) if N > 1 else torch.zeros(Mq, dtype=torch.bool, device="cuda")
            err = float(((idx != want) & ~near).sum()) + (0.0 if N <= 8 or int(idx[0]) == 1 + 7 else 1.0)
            tol = 0.0
            desc += f" rows={Mq} N={N} D={D}"
        elif kind == "split_gemm":
            # r06: x . w^T through split-bf16 operands (hi.hi + lo.hi + hi.lo over 3K) against an fp64 product
            K_, N_ = rng.choice([8, 24, 200, 1152]), rng.choice([8, 96, 257])
            Ms = min(M, 700)
            x = torch.randn(Ms, K_, device="cuda", generator=g) * rng.choice([0.1, 1.0, 20.0])
            w = torch.randn(N_, K_, device="cuda", generator=g) * K_ ** -0.5
            y = ops.gemm(ops.split3(x, 0), ops.split3(w, 1), out_fp32=True)
            ref = (x.double() @ w.double().t()).float()
            err, tol = float((y - ref).abs().max()) / (float(ref.abs().max()) + 1e-9), 5e-5
            desc += f" rows={Ms} N={N_} K={K_}"
        elif kind == "groupnorm_f32":
            Cg, HW, Bn = rng.choice([32, 64, 128, 512]), rng.choice([1, 7, 64, 256, 1000]), rng.choice([1, 2, 3])
            x = torch.randn(Bn * HW, Cg, device="cuda", generator=g) * rng.choice([0.3, 2.0]) + rng.choice([0.0, 1.0])
            wt, bs = 1 + 0.2 * torch.randn(Cg, device="cuda", generator=g), 0.1 * torch.randn(Cg, device="cuda", generator=g)
            sw = rng.random() < 0.5
            form = rng.choice(["f32", "bf16, fp32 params", "bf16, bf16 params", "bf16, in place"])
            if form == "f32":
                y = ops.groupnorm_f32(x, Bn, HW, 32, wt, bs, 1e-6, sw)
            else:                                                      # the decoder-side kernels: bf16 rows (the error is the output rounding), parameters in either format
                x = x.to(BF)
                if form != "bf16, fp32 params": wt, bs = wt.to(BF), bs.to(BF)
                xin = x.clone()
                y = ops.groupnorm(xin, Bn, HW, 32, wt, bs, 1e-6, sw, out=xin if form.endswith("in place") else None).float()
                x, wt, bs = x.float(), wt.float(), bs.float()
            xg = x.double().view(Bn, HW, 32, Cg // 32)                  # (by hand in fp64: F.group_norm refuses a single value per group, the kernel does not)
            mu, var = xg.mean((1, 3), keepdim=True), xg.var((1, 3), unbiased=False, keepdim=True)
            r_ = (((xg - mu) / (var + 1e-6).sqrt()).view(Bn * HW, Cg) * wt.double() + bs.double()).float()
            ref = r_ * torch.sigmoid(r_) if sw else r_
            err, tol = float((y - ref).abs().max()) / (float(ref.abs().max()) + 1e-9), (2e-5 if form == "f32" else 4.5e-3)
            desc += f" B={Bn} HW={HW} C={Cg} swish={sw} {form}"
        else:
            V = rng.choice([5, 320, 32017, 152064])
            Mv = min(M, 300)
            lg = torch.randn(Mv, V, device="cuda", generator=g)
            sup = rng.choice([-1, 0, V - 1, int(lg[0].argmax())])
            got = ops.argmax(lg, suppress=sup)
            l2 = lg.clone()
            if sup >= 0: l2[:, sup] = float("-inf")
            err, tol = float((got != l2.argmax(-1)).sum()), 0.0
            desc += f" V={V} rows={Mv} suppress={sup}"
        torch.cuda.synchronize()
    except _lib.CrabHipError as e:
        msg = str(e)
        if "error -1:" in msg or "error -3:" in msg:
            rejected += 1
            k = kind + ": " + msg.split(":", 2)[-1].strip()[:80]
            why[k] = why.get(k, 0) + 1
            continue
        bad.append(desc + " -> " + msg[:200]); continue
    done += 1
    if not (err <= tol): bad.append(desc + f" -> err {err:.3e} (tol {tol})")
    if not ok_guard: bad.append(desc + " -> a store landed outside the output block")
print(f"{done} cases computed, {rejected} rejected by the library, {len(bad)} failures")
for k_, v_ in sorted(why.items(), key=lambda kv: -kv[1]): print(f"  rejected x{v_}: {k_}")
for b_ in bad[:40]: print("FAIL", b_)
sys.exit(1 if bad else 0)
