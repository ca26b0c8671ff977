"""Differential fuzz of the pixel-task kernels through the C-ABI at random shapes (r06): the SegModule helpers of csrc/seg_ops.hip (3x3 im2col plain
and strided, the ConvTranspose pixel shuffle, bilinear resize in both input formats with accumulation, the random-Fourier positional encoding, the
row-broadcast add, the previous-mask gate, the group mean, the PNG label map, the in-place activation), the VQGAN helpers of csrc/vq_ops.hip that
fuzz_ops.py does not reach (nearest upsampling, the two row softmaxes, codebook norms, the bf16-path arg-min) and the eval loops' metrics of
csrc/seg_metrics.hip (mask_iou / metric_s_for_null / Eval_Fmeasure / calc_color_miou_fscore / color_mask_to_label) against torch and against
oracle/metrics_oracle.py (test infrastructure, pinned to the reference's own functions by tests/golden/seg_metrics.npz).  Copies, index work and
pixel counts must be EQUAL; floating-point values within the rounding of their output format.   python scripts/fuzz_seg.py [cases] [seed]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from crab_amd import ops, _lib, avss_utils as AU, harness
from oracle import metrics_oracle as MO

BF = torch.bfloat16
NCASE = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad, rejected, done, why, worst = [], 0, 0, {}, {}
KINDS = ["im2col", "im2col_strided", "pixel_shuffle", "bilinear", "dense_pe", "add_rows", "mask_gate", "group_mean", "mask_labels", "act",
         "upsample", "softmax", "sqnorm", "vq_argmin", "mask_iou", "fmeasure", "miou_fscore", "color_to_label"]


def relerr(a, b):
    return float((a.float() - b.float()).abs().max()) / (float(b.float().abs().max()) + 1e-9)


def tokens(x):                                        # [B, C, h, w] -> token-major [B*h*w, C]
    B, C, h, w = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * h * w, C).contiguous()


def close64(a, b, rel=1e-6):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    m = ~np.isnan(a)
    return bool(np.all(np.abs(a[m] - b[m]) <= rel * np.maximum(1.0, np.abs(b[m]))))


for case in range(NCASE):
    g = torch.Generator(device="cuda").manual_seed(case)
    kind = rng.choice(KINDS)
    desc, err, tol = f"case {case}: {kind}", 0.0, 0.0
    rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
    try:
        if kind in ("im2col", "im2col_strided"):
            B, h, w, C = rng.choice([1, 2, 3]), rng.choice([1, 2, 5, 14, 16, 33]), rng.choice([1, 3, 8, 14, 31]), rng.choice([8, 16, 64, 136])
            x = rn(B, C, h, w).to(BF)
            if kind == "im2col":
                got = ops.im2col3x3(tokens(x), B, h, w)
                cols = F.unfold(x.float(), 3, padding=1)                                      # [B, C*9, h*w], row index c*9 + tap
                oh, ow = h, w
            else:
                stride, pt, pl = rng.choice([1, 2]), rng.choice([0, 1]), rng.choice([0, 1])
                pb, pr = rng.choice([0, 1, 2]), rng.choice([0, 1, 2])
                oh, ow = (h + pt + pb - 3) // stride + 1, (w + pl + pr - 3) // stride + 1
                if oh < 1 or ow < 1:
                    continue
                got = ops.im2col3x3_strided(tokens(x), B, h, w, stride, pt, pl, oh, ow)
                cols = F.unfold(F.pad(x.float(), (pl, pr, pt, pb)), 3, stride=stride)
                desc += f" stride={stride} pad=({pt},{pl},{pb},{pr})"
            want = cols.view(B, C, 9, oh * ow).permute(0, 3, 2, 1).reshape(B * oh * ow, 9 * C)    # column (tap, c)
            err = float((got.float() != want).sum())
            desc += f" B={B} h={h} w={w} C={C}"
        elif kind == "pixel_shuffle":
            h, w, Co = rng.choice([1, 3, 14, 28]), rng.choice([1, 5, 14, 28]), rng.choice([1, 8, 20, 64])
            gm, bias = rn(h * w, 4 * Co).to(BF), (rn(Co).to(BF) if rng.random() < 0.7 else None)
            got = ops.pixel_shuffle2x(gm, bias, h, w, Co)
            v = gm.float().view(h, w, 2, 2, Co).permute(0, 2, 1, 3, 4).reshape(4 * h * w, Co)
            want = (v + (bias.float() if bias is not None else 0.0)).to(BF)
            err = float((got.view(torch.int16) != want.view(torch.int16)).sum())
            desc += f" h={h} w={w} Co={Co} bias={bias is not None}"
        elif kind == "bilinear":
            C, h, w = rng.choice([1, 3, 71]), rng.choice([1, 2, 7, 28, 56]), rng.choice([1, 3, 28, 56])
            H, W = rng.choice([1, h, 2 * h, 37, 224]), rng.choice([1, w, 4 * w, 50, 224])
            f32 = rng.random() < 0.5
            x = rn(C, h, w) if f32 else rn(C, h, w).to(BF)
            tm = rng.random() < 0.5                                                             # token-major storage [h*w, C]
            store = x.permute(1, 2, 0).contiguous() if tm else x.contiguous()
            strides = (1, w * C, C) if tm else (h * w, w, 1)
            alpha, beta = rng.choice([1.0, 0.5]), rng.choice([0.0, 0.0, 1.0, 0.25])
            out0 = rn(C, H, W)
            out = out0.clone()
            ops.bilinear(store, strides, C, h, w, out, alpha, beta)
            want = beta * out0 + alpha * F.interpolate(x.float()[None], size=(H, W), mode="bilinear", align_corners=False)[0]
            err, tol = relerr(out, want), 2e-6
            desc += f" C={C} {h}x{w}->{H}x{W} fp32={f32} token_major={tm} alpha={alpha} beta={beta}"
        elif kind == "dense_pe":
            h, w, Fq = rng.choice([1, 7, 28, 64]), rng.choice([1, 9, 28, 64]), rng.choice([1, 16, 128])
            G = rn(2, Fq)
            got = ops.dense_pe(G, h, w)
            ys, xs = (torch.arange(h, device="cuda").float() + 0.5) / h, (torch.arange(w, device="cuda").float() + 0.5) / w
            c = torch.stack([xs[None, :].expand(h, w), ys[:, None].expand(h, w)], -1)           # (x, y)
            a = 2 * np.pi * ((2 * c - 1).double() @ G.double())
            want = torch.cat([a.sin(), a.cos()], -1).reshape(h * w, 2 * Fq).float()
            err, tol = float((got.float() - want).abs().max()), 4.1e-3 + 2e-5 * float(a.abs().max())      # bf16 output + fp32 argument rounding of sin / cos
            desc += f" h={h} w={w} F={Fq}"
        elif kind == "add_rows":
            M, D, br = rng.choice([1, 5, 64, 300, 1024]), rng.choice([8, 64, 256, 1000]), rng.choice([1, 3, 64])
            a, b = rn(M, D + 8).to(BF)[:, :D], rn(br, D).to(BF)
            got = ops.add_rows(a, b)
            want = (a.float() + b.float()[torch.arange(M, device="cuda") % br]).to(BF)
            err = float((got.view(torch.int16) != want.view(torch.int16)).sum())
            desc += f" M={M} D={D} brows={br}"
        elif kind == "mask_gate":
            M, D, nc = rng.choice([1, 3, 4, 5, 784]), rng.choice([1, 32, 100, 256]), rng.choice([1, 2, 71, 130])
            prev, src = (rn(M, nc) * 3).to(BF), rn(M, D).to(BF)
            want = (src.float() * (torch.sigmoid(prev.float().mean(1, keepdim=True)) + 1)).to(BF)
            got = ops.mask_gate(prev, src.clone())
            err, tol = relerr(got, want), 8e-3
            desc += f" M={M} D={D} classes={nc}"
        elif kind == "group_mean":
            G_, T, D = rng.choice([1, 6, 71]), rng.choice([1, 2, 6, 10]), rng.choice([1, 8, 256, 4096])
            x = rn(G_ * T, D).to(BF)
            sc = rng.choice([1.0, 1.0 / T])
            got = ops.group_mean(x, G_, T, sc)
            want = (sc * x.float().view(G_, T, D).sum(1)).to(BF)
            err, tol = relerr(got, want), 8e-3
            desc += f" G={G_} T={T} D={D}"
        elif kind == "mask_labels":
            C, H, W = rng.choice([1, 2, 71, 255]), rng.choice([1, 7, 224]), rng.choice([1, 13, 224])
            pred = rn(C, H, W)
            mode = rng.choice(["plain", "ties", "nan", "zeros"])
            if mode == "ties": pred = pred.round()
            if mode == "zeros": pred = torch.zeros_like(pred)
            if mode == "nan": pred[rng.randrange(C), :, ::3] = float("nan")
            got = ops.mask_labels(pred)
            want = ((pred[0] > 0).to(torch.uint8) * 255) if C == 1 else pred.argmax(0).to(torch.uint8)      # sigmoid(x) > 0.5 <=> x > 0; NaN = the maximum, first wins
            err = float((got != want).sum())
            desc += f" C={C} {H}x{W} {mode}"
        elif kind == "act":
            n, act = rng.choice([1, 7, 256, 100003]), rng.choice(["gelu", "quick_gelu", "relu", "silu"])
            x = (rn(n) * 3).to(BF)
            xf = x.float()
            want = {"gelu": F.gelu(xf), "quick_gelu": xf * torch.sigmoid(1.702 * xf), "relu": F.relu(xf), "silu": F.silu(xf)}[act]
            got = ops.act_inplace(x.clone(), act)
            err, tol = float((got.float() - want).abs().max()) / (float(want.abs().max()) + 1e-9), 4.5e-3
            desc += f" n={n} {act}"
        elif kind == "upsample":
            B, h, w, C = rng.choice([1, 2]), rng.choice([1, 3, 16]), rng.choice([1, 5, 16]), rng.choice([8, 64, 136])
            x = rn(B, C, h, w).to(BF)
            got = ops.upsample_nearest2x(tokens(x), B, h, w)
            want = tokens(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"))
            err = float((got.float() != want).sum())
            desc += f" B={B} h={h} w={w} C={C}"
        elif kind == "softmax":
            M, N = rng.choice([1, 3, 256, 1000]), rng.choice([1, 2, 63, 256, 257, 4096])
            x = (rn(M, N + 3) * rng.choice([1.0, 30.0]))[:, :N]
            sc = rng.choice([1.0, 0.125])
            if rng.random() < 0.3: x[0, 0] = 3.0e4                                               # one dominant entry: exp of the rest underflows
            want = torch.softmax(sc * x.double(), -1).float()
            if rng.random() < 0.5:
                got, tol = ops.softmax_rows_f32(x, sc), 3e-6
            else:
                got, tol = ops.softmax_rows(x, sc).float(), 4.1e-3
            err = float((got - want).abs().max()) / (float(want.max()) + 1e-30)
            desc += f" M={M} N={N} scale={sc}"
        elif kind == "sqnorm":
            N, D = rng.choice([1, 3, 4, 5, 1000, 16384]), rng.choice([1, 4, 8, 256, 300])
            if rng.random() < 0.5:
                e = rn(N, D + 2)[:, :D]
                got = ops.row_sqnorm_f32(e)
            else:
                e = rn(N, D + 8).to(BF)[:, :D]
                got = ops.row_sqnorm(e)
            want = (e.double() ** 2).sum(1).float()
            err, tol = relerr(got, want), 2e-6
            desc += f" N={N} D={D} {e.dtype}"
        elif kind == "vq_argmin":
            M, N = rng.choice([1, 5, 256, 1000]), rng.choice([1, 2, 255, 256, 257, 16384])
            dots, e2 = rn(M, N + 1)[:, :N], rn(N).abs()
            if N > 4:
                dots[:, N - 1] = dots[:, 1]                                                      # a duplicated candidate: the first index wins
                e2[N - 1] = e2[1]
            got = ops.vq_argmin(dots, e2, offset=3)
            want = torch.argmin(e2[None, :] - 2 * dots, 1) + 3                                   # 2 * dots is exact: the same fp32 expression
            err = float((got != want).sum())
            desc += f" M={M} N={N}"
        elif kind in ("mask_iou", "fmeasure"):
            N, H, W = rng.choice([1, 2, 5]), rng.choice([1, 7, 32, 224]), rng.choice([1, 9, 32, 223, 224])
            pred = rn(N, H, W) * rng.choice([0.5, 4.0])
            gt = (rn(N, H, W) > rng.choice([-0.5, 0.0, 1.5])).float()
            mode = rng.choice(["plain", "empty gt", "full gt", "all negative", "all positive"])
            if mode == "empty gt": gt[rng.randrange(N)] = 0
            if mode == "full gt": gt[rng.randrange(N)] = 1
            if mode == "all negative": pred[rng.randrange(N)] = -pred[0].abs() - 1
            if mode == "all positive": pred[rng.randrange(N)] = pred[0].abs() + 1
            pn, gn = pred.cpu().numpy(), gt.cpu().numpy()
            desc += f" N={N} {H}x{W} {mode}"
            if kind == "mask_iou":
                iou, counts = AU.mask_iou(pred, gt, details=True)
                ok = np.array_equal(counts[:, :5].numpy(), MO.mask_counts(pn, gn)) and close64(iou.item(), MO.mask_iou(pn, gn))
                s = AU.metric_s_for_null(pred[:1])
                ok = ok and float(s.item()) == float(MO.metric_s_for_null(pn[:1]))
                if N == 1: ok = ok and float(iou.item()) == float(MO.mask_iou(pn, gn))          # one image: no sum involved, every operation rounded once
                err = 0.0 if ok else 1.0
            else:
                T = rng.choice([1, 2, 255, 1024])
                val, d = AU.Eval_Fmeasure(pred, gt, T, details=True)
                want, dw = MO.eval_fmeasure(pn, gn, T, th=AU.fmeasure_thresholds(T), details=True)
                ok = np.array_equal(d["ge"].numpy(), dw["ge"])
                ok = ok and d["images"] == dw["images"] and close64(val, want)
                err = 0.0 if ok else 1.0
                desc += f" T={T}"
        elif kind == "miou_fscore":
            BFr, C, H, W = rng.choice([1, 2, 10]), rng.choice([1, 2, 5, 71, 200]), rng.choice([1, 8, 31, 224]), rng.choice([1, 8, 30, 224])
            pred = rn(BFr, C, H, W)
            tgt = torch.randint(0, C, (BFr, H, W), device="cuda", generator=g)
            mode = rng.choice(["plain", "ignore 255", "negative", "ties", "one class"])
            if mode == "ignore 255": tgt[:, ::2, ::3] = 255
            if mode == "negative": tgt[0, :, ::2] = -1
            if mode == "ties": pred = pred.round()
            if mode == "one class": tgt[:] = C - 1
            ious, fs, cc, vid, d = AU.calc_color_miou_fscore(pred, tgt, details=True)
            pn, tn = pred.cpu().numpy(), tgt.cpu().numpy()
            mi, fw, cw, vw, iou_fc = MO.batch_miou_fscore(pn, tn)
            ok = np.array_equal(d["areas"].numpy(), MO.class_areas(pn, tn)) and np.array_equal(d["iou_fc"].numpy(), iou_fc)
            ok = ok and close64(ious.cpu().numpy(), mi) and close64(fs.cpu().numpy(), fw) and np.array_equal(cc.cpu().numpy(), cw)
            ok = ok and close64(torch.stack(vid).cpu().numpy(), vw)
            err = 0.0 if ok else 1.0
            desc += f" BF={BFr} C={C} {H}x{W} {mode}"
        else:
            n, H, W = rng.choice([1, 2, 71, 256]), rng.choice([1, 7, 224]), rng.choice([1, 5, 224])
            pal = harness.default_palette(max(n, 71))[:n] if n <= 71 else np.random.RandomState(case).randint(0, 256, (n, 3)).astype(np.uint8)
            idx = np.random.RandomState(case + 1).randint(0, n, (H, W))
            img = pal[idx].astype(np.uint8)
            if rng.random() < 0.5: img[::2, ::2] = (1, 2, 3)                                     # a colour outside the table
            got = harness.color_mask_to_label(img, pal).cpu().numpy()
            err = float((got != MO.color_mask_to_label(img, pal)).sum())
            desc += f" colours={n} {H}x{W}"
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
    n_, w_ = worst.get(kind, (0, 0.0))
    worst[kind] = (n_ + 1, max(w_, err))
    if not (err <= tol): bad.append(desc + f" -> err {err:.3e} (tol {tol})")
print("  per kind (cases, worst error): " + ", ".join(f"{k} {n} {w:.1e}" for k, (n, w) in sorted(worst.items())))
print(f"{done} cases computed, {rejected} rejected by the library, {len(bad)} failures")
for k_, v_ in sorted(why.items(), key=lambda kv: -kv[1]): print(f"  rejected x{v_}: {k_}")
for b_ in bad[:40]: print("FAIL", b_)
sys.exit(1 if bad else 0)
