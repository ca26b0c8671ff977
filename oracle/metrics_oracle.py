"""CPU restatement of the reference's segmentation metrics (SURVEY.md 8 f-1 anchors) -- TEST INFRASTRUCTURE ONLY (tests/, smoke());
the product path is crab_amd/avss_utils.py -> libcrab_hip.so (csrc/seg_metrics.hip).

Follows /root/reference utils/avss_utils.py: metric_s_for_null :8-19, mask_iou :22-47, _eval_pr :50-64, Eval_Fmeasure :67-96,
_batch_miou_fscore :379-419, calc_color_miou_fscore :422-435; and the final division of scripts/quick_start.py:437-447 (inference_avss).
numpy, fp32 arithmetic where the reference's tensors are fp32 (numpy does not fuse a*b + c), written loop by loop as the reference is.
Pinned by tests/golden/seg_metrics.npz (outputs of the reference's functions, make_golden.py metrics).

Two places where the reference's OWN value depends on the host it runs on, and what this restatement fixes them to:
  * `torch.sigmoid` in fp32 (a vectorised expf, <= 1 ulp from the correctly rounded value) -> sigmoid in fp64, rounded once to fp32.
    Only Eval_Fmeasure compares sigmoid values against thresholds; a pixel whose sigmoid sits within 1 ulp of a threshold may fall on the other
    side.  `sigmoid(x) > 0.5` (mask_iou, metric_s_for_null) is restated as x > 0: the two fp32 evaluations disagree only on x in (0, 2^-22).
  * `torch.linspace(0, 1 - 1e-10, 255)` in fp32 (vectorised base + i * step, last bit depends on the vector width) -> i / 254 rounded once from fp64
    (`thresholds()`); the fixture records torch's table so that tests can run either.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def sigmoid32(x: np.ndarray) -> np.ndarray:
    x64 = np.asarray(x, np.float64)
    with np.errstate(over="ignore"):
        return (1.0 / (1.0 + np.exp(-x64))).astype(F32)


def thresholds(num: int = 255) -> np.ndarray:
    """_eval_pr's thlist (avss_utils.py:56): linspace(0, fp32(1 - 1e-10) = 1.0, num)."""
    end = float(F32(1 - 1e-10))
    if num == 1:
        return np.zeros((1,), F32)
    return (np.arange(num, dtype=np.float64) * (end / (num - 1))).astype(F32)


def metric_s_for_null(pred: np.ndarray) -> np.float32:
    """avss_utils.py:8-19: sqrt(sum(sigmoid(pred) > 0.5) / numel); int64 sum / int -> fp32 true division."""
    assert pred.ndim == 3
    x = int((np.asarray(pred, F32) > 0).sum())
    return np.sqrt(F32(x) / F32(pred.size), dtype=F32)


def mask_counts(pred: np.ndarray, target: np.ndarray) -> np.ndarray:
    """[N][5] = {pred, target, pred & target, pred | target, !pred & !target} pixel counts (the integer content of mask_iou)."""
    p = np.asarray(pred, F32) > 0
    t = np.asarray(target) != 0
    ax = (1, 2)
    return np.stack([p.sum(ax), t.sum(ax), (p & t).sum(ax), (p | t).sum(ax), (~p & ~t).sum(ax)], 1).astype(np.int64)


def mask_iou(pred: np.ndarray, target: np.ndarray, eps: float = 1e-7) -> np.float32:
    """avss_utils.py:22-47."""
    assert pred.ndim == 3 and pred.shape == target.shape
    N = pred.shape[0]
    num_pixels = pred.shape[-1] * pred.shape[-2]
    c = mask_counts(pred, target)
    inter, union = c[:, 2].astype(F32), c[:, 3].astype(F32)
    no_obj = c[:, 1] == 0
    inter[no_obj] = c[no_obj, 4].astype(F32)
    union[no_obj] = F32(num_pixels)
    acc = F32(0)
    for n in range(N):                                          # torch.sum over N fp32 values: sequential here (tests allow 1e-6 relative)
        acc = F32(acc + F32(inter[n] / F32(union[n] + F32(eps))))
    return F32(acc / F32(N))


def eval_pr(y_pred: np.ndarray, y: np.ndarray, th: np.ndarray):
    """avss_utils.py:50-64, one threshold after the other as the reference does; also returns the integer counts."""
    num = len(th)
    prec, recall = np.zeros(num, F32), np.zeros(num, F32)
    tpc, cntc = np.zeros(num, np.int64), np.zeros(num, np.int64)
    ysum = F32((y != 0).sum())
    for i in range(num):
        y_temp = y_pred >= th[i]
        tp = int((y_temp & (y != 0)).sum())
        cnt = int(y_temp.sum())
        tpc[i], cntc[i] = tp, cnt
        prec[i] = F32(tp) / F32(F32(cnt) + F32(1e-20))
        recall[i] = F32(tp) / F32(ysum + F32(1e-20))
    return prec, recall, tpc, cntc


def eval_fmeasure(pred: np.ndarray, gt: np.ndarray, pr_num: int = 255, th: np.ndarray = None, details: bool = False):
    """avss_utils.py:67-96."""
    th = thresholds(pr_num) if th is None else np.asarray(th, F32)
    sp = sigmoid32(pred)
    N = pred.shape[0]
    beta2 = 0.3
    avg_f, img_num = None, 0
    score = np.zeros(pr_num, F32)
    ge = np.zeros((N, 2, pr_num), np.int64)
    fs_all = np.zeros((N, pr_num), F32)
    for n in range(N):
        prec, recall, tpc, cntc = eval_pr(sp[n], gt[n], th)
        ge[n, 0], ge[n, 1] = tpc, cntc
        with np.errstate(divide="ignore", invalid="ignore"):
            f = F32(1 + beta2) * prec * recall / (F32(beta2) * prec + recall)
        f = f.astype(F32)
        f[np.isnan(f)] = 0
        fs_all[n] = f
        if float(np.mean(gt[n])) == 0.0:                        # totally black ground truth: out of consideration (:84-85)
            continue
        avg_f = f.copy() if avg_f is None else (avg_f + f).astype(F32)
        img_num += 1
        score = (avg_f / F32(img_num)).astype(F32)
    val = float(score.max())
    if details:
        return val, {"ge": ge, "fscore": fs_all, "score": score, "images": img_num}
    return val


def class_areas(pred: np.ndarray, target: np.ndarray) -> np.ndarray:
    """[BF][3][C] = the three histc calls of _batch_miou_fscore (avss_utils.py:386-402): {inter, pred, lab} class areas.  The reference shifts
    both maps by one, zeroes the prediction where target + 1 <= 0 and histograms the values in [1, nclass]."""
    BF, C = pred.shape[0], pred.shape[1]
    predict = np.argmax(np.asarray(pred, F32), 1) + 1           # argmax of the softmax = argmax of the logits (first maximum)
    tgt = np.asarray(target, np.int64) + 1
    predict = predict * (tgt > 0)
    inter = predict * (predict == tgt)
    out = np.zeros((BF, 3, C), np.int64)
    for f in range(BF):
        for k, m in enumerate((inter[f], predict[f], tgt[f])):
            v = m[(m >= 1) & (m <= C)]
            out[f, k] = np.bincount(v - 1, minlength=C)[:C]
    return out


def batch_miou_fscore(pred: np.ndarray, target: np.ndarray, beta2: float = 0.3):
    """avss_utils.py:379-419 -> (ious [C], fscores [C], cls_count [C], vid_miou [BF], iou_fc [BF, C])."""
    areas = class_areas(pred, target)
    BF, _, C = areas.shape
    ious, fscores, cls_count = np.zeros(C, F32), np.zeros(C, F32), np.zeros(C, F32)
    vid, iou_fc = np.zeros(BF, F32), np.zeros((BF, C), F32)
    for f in range(BF):
        ai, ap, al = (areas[f, k].astype(F32) for k in range(3))
        au = (ap + al - ai).astype(F32)
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = (ai / (F32(2.220446049250313e-16) + au)).astype(F32)
            precision, recall = ai / ap, ai / al
            fscore = (F32(1 + beta2) * precision * recall / (F32(beta2) * precision + recall)).astype(F32)
        fscore[np.isnan(fscore)] = 0
        ious = (ious + iou).astype(F32)
        fscores = (fscores + fscore).astype(F32)
        cls_count[au != 0] += 1
        iou_fc[f] = iou
        s = F32(0)
        for c in range(C):                                      # torch.sum over C fp32 values: sequential here (tests allow 1e-6 relative)
            s = F32(s + iou[c])
        with np.errstate(divide="ignore", invalid="ignore"):
            vid[f] = s / F32((iou != 0).sum())
    return ious, fscores, cls_count, vid, iou_fc


def avss_final(miou_pc: np.ndarray, fs_pc: np.ndarray, cls_pc: np.ndarray) -> dict:
    """scripts/quick_start.py:437-447: per-class sums / counts, NaN -> 0, mean over all classes and over all but the last."""
    with np.errstate(divide="ignore", invalid="ignore"):
        mi = (np.asarray(miou_pc, F32) / np.asarray(cls_pc, F32)).astype(F32)
        fs = (np.asarray(fs_pc, F32) / np.asarray(cls_pc, F32)).astype(F32)
    mi[np.isnan(mi)] = 0
    fs[np.isnan(fs)] = 0
    return {"miou": float(mi.mean(dtype=F32)), "miou_noBg": float(mi[:-1].mean(dtype=F32)),
            "f_score": float(fs.mean(dtype=F32)), "f_score_noBg": float(fs[:-1].mean(dtype=F32))}


def get_v2_pallete(num_cls: int = 71) -> np.ndarray:
    """dataset/quick_start_dataset.py:35-59 (_getpallete): the PASCAL-VOC bit shuffle, [num_cls, 3]."""
    pal = np.zeros((num_cls, 3), np.int64)
    for j in range(num_cls):
        lab, i = j, 0
        while lab > 0:
            pal[j, 0] |= ((lab >> 0) & 1) << (7 - i)
            pal[j, 1] |= ((lab >> 1) & 1) << (7 - i)
            pal[j, 2] |= ((lab >> 2) & 1) << (7 - i)
            i += 1
            lab >>= 3
    return pal


def color_mask_to_label(mask: np.ndarray, v_pallete: np.ndarray) -> np.ndarray:
    """dataset/quick_start_dataset.py:63-73: one equality plane per colour, argmax over the planes (first match; 0 when no colour matches)."""
    mask_array = np.asarray(mask).astype("int32")
    semantic_map = []
    for colour in v_pallete:
        semantic_map.append(np.all(np.equal(mask_array, colour), axis=-1))
    return np.argmax(np.stack(semantic_map, axis=-1).astype(np.float32), axis=-1)
