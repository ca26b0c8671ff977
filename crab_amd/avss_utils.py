"""Segmentation metrics of the pixel-task eval loops on the device: the counterpart of the reference's utils/avss_utils.py
(metric_s_for_null :8-19, mask_iou :22-47, _eval_pr / Eval_Fmeasure :50-96, _batch_miou_fscore / calc_color_miou_fscore :379-435), with
the reference's names, argument meaning and return shapes, so that the loops of scripts/quick_start.py:52-450 read the same:

    iou = mask_iou(pred=pred_mask, target=gt_mask)                # reference: pred_mask.cpu(), gt_mask.cpu()
    fscore = Eval_Fmeasure(pred=pred_mask, gt=gt_mask)
    s = metric_s_for_null(pred_mask)
    miou_pc, fscore_pc, cls_pc, _ = calc_color_miou_fscore(pred=pred_mask.unsqueeze(0), target=gt_mask, T=1)

The reference moves every predicted mask to the host first; here the masks stay on the device they were produced on and every function is
one counting pass + one finishing launch of libcrab_hip.so (crab_mask_iou / crab_fmeasure / crab_miou_fscore, csrc/seg_metrics.hip).
Host tensors are refused (CrabHipError): there is no CPU path.  Pixel counts are exact integers; the fp32 ratios follow the reference's
operation order.  Ground-truth masks must be binary ({0, 1}, as the reference's datasets build them: quick_start_dataset.py:464-468) -
anything else raises.

`details=True` on a function additionally returns the integer counts behind the value (tests, reports)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib
from .ops import _dev, _p, _stream

_TH: Dict[Tuple[int, int], torch.Tensor] = {}


def fmeasure_thresholds(pr_num: int = 255) -> np.ndarray:
    """The thresholds of _eval_pr: the reference's own expression, `torch.linspace(0, 1 - 1e-10, num)` (avss_utils.py:56), evaluated by the same
    library on the host (fp32; 1 - 1e-10 rounds to 1.0f) and uploaded once per device.  ADVICE r05: a table rounded from fp64 (r05) differed from
    torch's tensor by one ulp on 10 of the 255 entries - a pixel whose sigmoid falls into such a gap would flip one `>=` count.  What remains: the
    kernel's sigmoid is 1 / (1 + exp(-x)) with the device's expf where the reference calls torch.sigmoid on its host (both fp32, each within
    an ulp or two of the true value), so a pixel whose sigmoid lies within ~2 ulp of a threshold may be counted differently; on the
    reference-generated fixture every count is equal (tests/test_seg_metrics.py)."""
    return torch.linspace(0, 1 - 1e-10, pr_num).numpy().astype(np.float32)


def _thresholds(device: torch.device, pr_num: int) -> torch.Tensor:
    key = (device.index or 0, pr_num)
    if key not in _TH:
        _TH[key] = torch.from_numpy(fmeasure_thresholds(pr_num)).to(device)
    return _TH[key]


def _planes(x: torch.Tensor, what: str) -> torch.Tensor:
    if x.dim() != 3:
        raise _lib.CrabHipError(f"{what}: [N, H, W] expected, got {tuple(x.shape)}")
    return x.float().contiguous()


def _binary_counts(pred: torch.Tensor, target):
    d = _dev(pred)
    p = _planes(pred, "pred")
    N, H, W = p.shape
    t = None
    if target is not None:
        if tuple(target.shape) != tuple(pred.shape):
            raise _lib.CrabHipError(f"pred {tuple(pred.shape)} and target {tuple(target.shape)} differ")
        if _dev(target) != d:
            raise _lib.CrabHipError("pred and target live on different devices")
        t = _planes(target, "target")
    counts = torch.empty((N, 6), device=p.device, dtype=torch.int32)
    out = torch.empty((2,), device=p.device, dtype=torch.float32)
    return d, p, t, counts, out, N, H * W


def metric_s_for_null(pred: torch.Tensor, details: bool = False):
    """sqrt(#{sigmoid(pred) > 0.5} / #pixels) of a [1, h, w] mask predicted for a null reference (avss_utils.py:8-19) -> 0-dim fp32 tensor."""
    d, p, _t, counts, out, N, hw = _binary_counts(pred, None)
    _lib.check(_lib.load().crab_mask_iou(_lib.ctx(d), _stream(), _p(p), None, N, hw, 0.0, _p(counts), _p(out)), d)
    return (out[1], counts) if details else out[1]


def mask_iou(pred: torch.Tensor, target: torch.Tensor, eps: float = 1e-7, size_average: bool = True, details: bool = False):
    """Mean over the N masks of |pred & target| / (|pred | target| + eps), an empty target scoring its correctly empty pixels over all pixels
    (avss_utils.py:22-47; `size_average` is accepted and ignored, as there) -> 0-dim fp32 tensor.  pred [N, H, W] logits, target [N, H, W] in {0, 1}."""
    d, p, t, counts, out, N, hw = _binary_counts(pred, target)
    _lib.check(_lib.load().crab_mask_iou(_lib.ctx(d), _stream(), _p(p), _p(t), N, hw, float(eps), _p(counts), _p(out)), d)
    c = counts.cpu()                                       # (the reference's callers read the value with .item() right away: one sync either way)
    if int(c[:, 5].sum()):
        raise _lib.CrabHipError("mask_iou: target must be binary ({0, 1})")
    return (out[0], c) if details else out[0]


def Eval_Fmeasure(pred: torch.Tensor, gt: torch.Tensor, pr_num: int = 255, details: bool = False):
    """max over the pr_num thresholds of the F-measure (beta^2 = 0.3) averaged over the images with a non-empty ground truth (avss_utils.py:67-96)
    -> Python float, like the reference's `.item()`."""
    d = _dev(pred)
    if tuple(gt.shape) != tuple(pred.shape):
        raise _lib.CrabHipError(f"pred {tuple(pred.shape)} and gt {tuple(gt.shape)} differ")
    if _dev(gt) != d:
        raise _lib.CrabHipError("pred and gt live on different devices")
    p, g = _planes(pred, "pred"), _planes(gt, "gt")
    N, H, W = p.shape
    T = int(pr_num)
    if not 1 <= T <= 1024:
        raise _lib.CrabHipError("Eval_Fmeasure: 1 <= pr_num <= 1024")
    th = _thresholds(p.device, T)
    e = lambda *s, dt: torch.empty(s, device=p.device, dtype=dt)
    ge, ysum = e(N, 2, T, dt=torch.int32), e(N, 2, dt=torch.int32)
    fscore, score, best = e(N, T, dt=torch.float32), e(T, dt=torch.float32), e(2, dt=torch.float32)
    _lib.check(_lib.load().crab_fmeasure(_lib.ctx(d), _stream(), _p(p), _p(g), N, H * W, _p(th), T, 0.3, _p(ge), _p(ysum), _p(fscore), _p(score),
                                         _p(best)), d)
    ys = ysum.cpu()
    if int(ys[:, 1].sum()):
        raise _lib.CrabHipError("Eval_Fmeasure: gt must be binary ({0, 1})")
    val = float(best[0].item())
    if details:
        return val, {"ge": ge.cpu(), "ysum": ys, "fscore": fscore.cpu(), "score": score.cpu(), "images": int(best[1].item())}
    return val


def calc_color_miou_fscore(pred: torch.Tensor, target: torch.Tensor, T: int = 10, details: bool = False):
    """The AVSS J / F sums of one batch (avss_utils.py:379-435): pred [BF, C, H, W] class logits (background included), target [BF, H, W] integer
    class ids -> (ious [C], fscores [C], cls_count [C], vid_miou_list: BF 0-dim tensors), fp32 on the device.  The caller accumulates the three
    vectors over the dataset and divides at the end (scripts/quick_start.py:399-447).  `T` is accepted and unused, as in the reference."""
    d = _dev(pred)
    if pred.dim() != 4 or target.dim() != 3 or tuple(target.shape) != (pred.shape[0],) + tuple(pred.shape[2:]):
        raise _lib.CrabHipError(f"calc_color_miou_fscore: pred [BF, C, H, W] and target [BF, H, W] expected, got {tuple(pred.shape)} / {tuple(target.shape)}")
    if _dev(target) != d:
        raise _lib.CrabHipError("pred and target live on different devices")
    if target.dtype.is_floating_point or target.dtype == torch.bool:
        raise _lib.CrabHipError("calc_color_miou_fscore: integer class ids expected")
    p = pred.float().contiguous()
    t = target.to(torch.int64).contiguous()
    BF, C_, H, W = p.shape
    if not 1 <= C_ <= 1024:
        raise _lib.CrabHipError("calc_color_miou_fscore: 1 <= classes <= 1024")
    e = lambda *s, dt=torch.float32: torch.empty(s, device=p.device, dtype=dt)
    areas, iou_fc = e(BF, 3, C_, dt=torch.int32), e(BF, C_)
    ious, fscores, cls_count, vid = e(C_), e(C_), e(C_), e(BF)
    _lib.check(_lib.load().crab_miou_fscore(_lib.ctx(d), _stream(), _p(p), _p(t), BF, C_, H * W, 0.3, _p(areas), _p(iou_fc), _p(ious), _p(fscores),
                                            _p(cls_count), _p(vid)), d)
    vid_list: List[torch.Tensor] = list(vid.unbind(0))
    if details:
        return ious, fscores, cls_count, vid_list, {"areas": areas.cpu(), "iou_fc": iou_fc.cpu()}
    return ious, fscores, cls_count, vid_list


class AVSSMeter:
    """The running sums of the reference's inference_avss loop and its final division (scripts/quick_start.py:364-371, 399-447): per-class IoU and
    F sums and the number of frames each class appeared in; `result()` = mean over ALL classes of sum / count with 0 / 0 -> 0, with and without
    the last class.  The sums are fp32 device tensors; nothing syncs until result()."""

    def __init__(self, n_classes: int = 71, device="cuda"):
        z = lambda: torch.zeros((n_classes,), device=device, dtype=torch.float32)
        self.miou_pc, self.fs_pc, self.cls_pc = z(), z(), z()

    def update(self, pred_mask: torch.Tensor, gt_mask: torch.Tensor):
        """pred_mask [C, H, W] (one sample's class logits), gt_mask [1, H, W] class ids (quick_start.py:395-403)."""
        i, f, c, _ = calc_color_miou_fscore(pred=pred_mask.unsqueeze(0), target=gt_mask, T=1)
        self.miou_pc += i
        self.fs_pc += f
        self.cls_pc += c

    def result(self) -> Dict[str, float]:
        m, f, c = (x.cpu().numpy() for x in (self.miou_pc, self.fs_pc, self.cls_pc))
        with np.errstate(divide="ignore", invalid="ignore"):
            mi, fs = m / c, f / c
        mi[np.isnan(mi)] = 0
        fs[np.isnan(fs)] = 0
        return {"miou": float(mi.mean(dtype=np.float32)), "miou_noBg": float(mi[:-1].mean(dtype=np.float32)),
                "f_score": float(fs.mean(dtype=np.float32)), "f_score_noBg": float(fs[:-1].mean(dtype=np.float32))}
